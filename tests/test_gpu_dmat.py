"""The multi-GPU schedules behind the C boundary (include/m4ri_amd.h part 4, m4ri_amd/csrc/multi.hip second half): distributed,
device-resident matrices (m4ri_amd_dmat) and m4ri_amd_dmat_mul -- the reference's multi-core entry is a C function
(mzd_mul_mp, m4ri/mp.c:158-297), so this is where the schedules live.  A one-GPU box runs them with several "virtual" ranks on
device 0 (own streams, buffers, host threads; peer copies device 0 -> device 0): the code a node of 8 GPUs runs, minus the
links.  Everything is checked bit for bit against the oracle, the BASELINE sizes against the reference's SHA-256."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch  # noqa: F401 -- before libm4ri_amd.so: the process gets ONE HIP runtime, the one torch ships

import m4ri_amd
from m4ri_amd import Dmat
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYOUTS = [m4ri_amd.LAYOUT_ROWS, m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.LAYOUT_CYCLIC2, m4ri_amd.LAYOUT_REPLICATED]


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)
    yield
    m4ri_amd.set_devices([])
    m4ri_amd.set_multi_variant(0)
    m4ri_amd.set_multi_threshold(16384)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_upload_download_convert_every_layout(world):
    m4ri_amd.set_devices([0] * world)
    for (r, c) in [(1000, 777), (1, 1), (513, 64), (7, 1300), (2048, 2048)]:
        M = Mzd.random(r, c, 5 + r)
        mats = {}
        for lay in LAYOUTS:
            d = Dmat(r, c, lay).upload(M)
            assert d.download().equal(M), (world, r, c, lay)
            info = d.info()
            assert (info.rows, info.ncols, info.layout, info.world, info.alive) == (r, c, lay, world, 1) and info.stride * 64 >= c and info.stride % 4 == 0
            mats[lay] = d
        for src in LAYOUTS:          # dst <- src for every pair (ROWS -> REPLICATED is the all-gather)
            for dst in LAYOUTS:
                e = Dmat(r, c, dst).convert_from(mats[src])
                assert e.download().equal(M), (world, r, c, src, dst)
                e.free()
        for d in mats.values():
            d.free()


def test_fill_is_the_global_fill_and_download_keeps_a_windows_parent():
    m4ri_amd.set_devices([0, 0, 0])
    for lay in LAYOUTS:
        d = Dmat(700, 450, lay).fill(77)
        assert d.download().equal(Mzd.random(700, 450, 77)), lay
        P = Mzd.random(1024, 1024, 9)
        want = Mzd(1024, 1024, buf=P.buf.copy())
        d.download(P.window(5, 128, 705, 578))          # 450 columns: a ragged last word inside the parent
        src, ww = Mzd.random(700, 450, 77), want.window(5, 128, 705, 578)
        mask = np.uint64(ww.high_bitmask)
        ww.valid_words()[:, :-1] = src.valid_words()[:, :-1]
        ww.valid_words()[:, -1] = (ww.valid_words()[:, -1] & ~mask) | (src.valid_words()[:, -1] & mask)
        assert np.array_equal(P.buf, want.buf), lay
        d.free()


SHAPES = [(64, 128, 128), (100, 256, 256), (77, 130, 65), (203, 300, 257), (1025, 1025, 1025), (2048, 2048, 4096), (1, 64, 64), (5, 1, 3),
          (1710, 1290, 1000), (4096, 3528, 4096), (3000, 512, 3000)]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("mode", ["slabs", "slabs_replicated_b", "strassen1", "strassen2", "mixed_layouts"])
def test_dmat_mul_vs_oracle(oracle, world, mode):
    m4ri_amd.set_devices([0] * world)
    la, lb, variant = {"slabs": (m4ri_amd.LAYOUT_ROWS, m4ri_amd.LAYOUT_ROWS, m4ri_amd.VARIANT_SLABS),
                       "slabs_replicated_b": (m4ri_amd.LAYOUT_ROWS, m4ri_amd.LAYOUT_REPLICATED, m4ri_amd.VARIANT_SLABS),
                       "strassen1": (m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.VARIANT_STRASSEN),
                       "strassen2": (m4ri_amd.LAYOUT_CYCLIC2, m4ri_amd.LAYOUT_CYCLIC2, m4ri_amd.VARIANT_STRASSEN),
                       "mixed_layouts": (m4ri_amd.LAYOUT_CYCLIC2, m4ri_amd.LAYOUT_ROWS, m4ri_amd.VARIANT_SLABS)}[mode]
    for (m, l, n) in SHAPES:
        A, B, C0 = Mzd.random(m, l, 11), Mzd.random(l, n, 12), Mzd.random(m, n, 13)
        dA, dB = Dmat(m, l, la).upload(A), Dmat(l, n, lb).upload(B)
        dC = Dmat(m, n, la).upload(C0)                       # dirty C is overwritten
        m4ri_amd.dmat_mul(dC, dA, dB, False, 0, m4ri_amd.VARIANT_AUTO if mode != "mixed_layouts" else variant)
        st = m4ri_amd.multi_stats()
        assert st.variant == variant and st.world == world and (st.converted > 0) == (mode == "mixed_layouts"), (mode, st.variant, st.converted)
        if variant == m4ri_amd.VARIANT_STRASSEN:
            assert st.levels == (1 if mode == "strassen1" else 2) and st.sub_products == 7 ** st.levels
        want = oracle.mul(None, A, B, 0)
        assert dC.download().equal(want), (world, mode, m, l, n)
        m4ri_amd.dmat_mul(dC.upload(C0), dA, dB, True, 0, variant)
        assert dC.download().equal(oracle.addmul(C0.copy(), A, B, 0)), ("addmul", world, mode, m, l, n)
        for d in (dA, dB, dC):
            d.free()


def test_products_chain_on_the_devices(oracle):
    """(A*B)*E and F*(A*B) without a byte over PCIe in between: the result of one product is an operand of the next, in the layout
    it was left in, for both schedules; squares of a matrix (A == B) included."""
    m4ri_amd.set_devices([0] * 4)
    m, l, n, k = 1500, 1100, 1300, 900
    A, B, E, F = Mzd.random(m, l, 1), Mzd.random(l, n, 2), Mzd.random(n, k, 3), Mzd.random(k, m, 4)
    AB = oracle.mul(None, A, B, 0)
    for lay in (m4ri_amd.LAYOUT_ROWS, m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.LAYOUT_CYCLIC2):
        dA, dB, dE, dF = (Dmat(x.nrows, x.ncols, lay).upload(x) for x in (A, B, E, F))
        dAB, dABE, dFAB = Dmat(m, n, lay), Dmat(m, k, lay), Dmat(k, n, lay)
        m4ri_amd.dmat_mul(dAB, dA, dB)
        m4ri_amd.dmat_mul(dABE, dAB, dE)
        m4ri_amd.dmat_mul(dFAB, dF, dAB)
        assert dABE.download().equal(oracle.mul(None, AB, E, 0)) and dFAB.download().equal(oracle.mul(None, F, AB, 0)), lay
        S = Mzd.random(1200, 1200, 6)
        dS, dSS = Dmat(1200, 1200, lay).upload(S), Dmat(1200, 1200, lay)
        m4ri_amd.dmat_mul(dSS, dS, dS)
        assert dSS.download().equal(oracle.mul(None, S, S, 0)), lay


def test_schedule_choice_is_made_behind_the_c_boundary():
    """m4ri_amd_mul_multi / mzd_mul_mp pick the schedule by shape and world size (sharding.default_variant's rule, now in C): a short
    inner dimension takes row slabs at every world size (BASELINE.json configs[4]'s shape class), cubes take the Strassen sub-products
    from 5 ranks on, and the stats say which ran."""
    m4ri_amd.set_devices([0] * 8)
    for (m, l, n, variant) in [(16384, 16384, 16384, m4ri_amd.VARIANT_STRASSEN), (16384, 2048, 16384, m4ri_amd.VARIANT_SLABS)]:
        A, B = Mzd.random(m, l, 51), Mzd.random(l, n, 52)
        ref = m4ri_amd.mzd_mul(None, A, B, 0)
        assert m4ri_amd.mul_multi(Mzd.init(m, n), A, B, False, 0, 0).equal(ref)
        st = m4ri_amd.multi_stats()
        assert (st.variant, st.world, st.converted) == (variant, 8, 0), (m, l, n, st.variant)
        assert st.variant == m4ri_amd.multi_default_variant(8, m, l, n)
    m4ri_amd.set_devices([0] * 4)
    A, B = Mzd.random(16384, 16384, 51), Mzd.random(16384, 16384, 52)
    ref = m4ri_amd.mzd_mul(None, A, B, 0)
    assert m4ri_amd.mul_multi(Mzd.init(16384, 16384), A, B, False, 0, 0).equal(ref) and m4ri_amd.multi_stats().variant == m4ri_amd.VARIANT_SLABS
    assert m4ri_amd.multi_stats().overlap == 1            # 4 ranks, slab boundaries on words: the gather runs under the first product
    old = m4ri_amd.set_multi_variant(m4ri_amd.VARIANT_STRASSEN)
    try:
        assert m4ri_amd.mul_multi(Mzd.init(16384, 16384), A, B, False, 0, 0).equal(ref) and m4ri_amd.multi_stats().variant == m4ri_amd.VARIANT_STRASSEN
    finally:
        m4ri_amd.set_multi_variant(old)


def test_row_chunks_and_the_timeline():
    """The Strassen schedule in two row chunks per sub-product (operands of chunk 1 and results of chunk 0 travel under the
    multiplications): same bits, and the per-phase marks of every rank are ordered -- down < operands in < product done ... <
    slabs back < done."""
    m4ri_amd.set_devices([0] * 8)
    n = 32768
    dA, dB, dC = Dmat(n, n, m4ri_amd.LAYOUT_CYCLIC1).fill(3), Dmat(n, n, m4ri_amd.LAYOUT_CYCLIC1).fill(4), Dmat(n, n, m4ri_amd.LAYOUT_CYCLIC1)
    m4ri_amd.dmat_mul(dC, dA, dB)
    st = m4ri_amd.multi_stats()
    assert (st.variant, st.levels, st.sub_products, st.chunks) == (m4ri_amd.VARIANT_STRASSEN, 1, 7, 2)
    assert st.link_bytes == 3 * 7 * 7 * (n // 2 // 8) * (n // 2 // 8)   # 3 sides x 7 products x 7 remote slabs of (n/2/8) rows x n/2 bits
    for r in range(8):
        marks = m4ri_amd.multi_timeline(r)
        units = 2 if r < 7 else 0
        assert len(marks) == 1 + 2 * units + 2 and all(t >= 0 for t in marks), (r, marks)
        assert marks[0] <= marks[-2] <= marks[-1] and all(marks[1 + 2 * u] <= marks[2 + 2 * u] for u in range(units)), (r, marks)
    A = torch.empty((n, n // 64), dtype=torch.int64, device="cuda")
    B, C = torch.empty_like(A), torch.empty_like(A)
    m4ri_amd.fill_dev(A.data_ptr(), n // 64, n, n, 3)
    m4ri_amd.fill_dev(B.data_ptr(), n // 64, n, n, 4)
    m4ri_amd.mul_dev(C.data_ptr(), n // 64, A.data_ptr(), n // 64, B.data_ptr(), n // 64, n, n, n, False, 0)
    torch.cuda.synchronize()
    got = dC.download()
    assert np.array_equal(got.valid_words().view(np.int64), C.cpu().numpy())


def _golden(m, l, n, seeds):
    path = os.path.join(ROOT, "tests", "golden", "sha256.json")
    want = [e["sha256"] for e in json.load(open(path)) if (e["op"], e["m"], e["l"], e["n"], e["seed_a"], e["seed_b"]) == ("mul", m, l, n, *seeds)]
    assert want, "no golden SHA-256 for this product"
    return want[0]


@pytest.mark.parametrize("m,l,n,seeds,variant", [(65536, 65536, 65536, (3, 4), m4ri_amd.VARIANT_STRASSEN),      # BASELINE.json configs[3]
                                                 (131072, 8192, 131072, (5, 6), m4ri_amd.VARIANT_SLABS)])       # BASELINE.json configs[4]
def test_baseline_sizes_from_resident_operands_vs_reference_sha256(m, l, n, seeds, variant):
    """The two 8-GPU configurations of BASELINE.json at full size on 8 ranks from RESIDENT distributed operands (filled on the
    devices in the layout the automatic schedule wants), the downloaded C against the SHA-256 of the real reference's product."""
    m4ri_amd.set_devices([0] * 8)
    assert m4ri_amd.multi_default_variant(8, m, l, n) == variant
    lay = m4ri_amd.multi_layout_for(0, 8, m, l, n)
    assert lay == (m4ri_amd.LAYOUT_CYCLIC2 if variant == m4ri_amd.VARIANT_STRASSEN else m4ri_amd.LAYOUT_ROWS)   # 47 sub-products over 8 ranks
    dA, dB, dC = Dmat(m, l, lay).fill(seeds[0]), Dmat(l, n, lay).fill(seeds[1]), Dmat(m, n, lay)
    m4ri_amd.dmat_mul(dC, dA, dB)
    m4ri_amd.dmat_mul(dC, dA, dB)          # twice back to back: buffers and events are reused across operations
    m4ri_amd.multi_sync()
    st = m4ri_amd.multi_stats()
    assert (st.variant, st.world, st.converted) == (variant, 8, 0)
    if variant == m4ri_amd.VARIANT_STRASSEN:   # two levels as ONE application of the rank-47 scheme, a rank's 6 sub-products in batched products
        assert (st.levels, st.sub_products, st.chunks) == (2, 47, 1) and st.group >= 2, (st.levels, st.sub_products, st.chunks, st.group)
    C = dC.download()
    assert hashlib.sha256(C.masked().tobytes()).hexdigest() == _golden(m, l, n, seeds)
    for d in (dA, dB, dC):
        d.free()
    m4ri_amd.lib().m4ri_amd_release_workspace()


def test_config3_on_4_ranks_takes_the_47_way_split_vs_reference_sha256():
    """65536^3 on FOUR ranks: the automatic schedule is the 47-way split too (12 sub-products of 16384^3 per rank in batched products:
    6.58 ms per rank against the row slab's 8.00, profiles/r06_rank_batch_timing.log), below that size and on 2 ranks the row slabs."""
    m = l = n = 65536
    m4ri_amd.set_devices([0] * 4)
    assert m4ri_amd.multi_default_variant(4, m, l, n) == m4ri_amd.VARIANT_STRASSEN and m4ri_amd.multi_default_variant(2, m, l, n) == m4ri_amd.VARIANT_SLABS
    assert m4ri_amd.multi_default_variant(4, 32768, 32768, 32768) == m4ri_amd.VARIANT_SLABS
    lay = m4ri_amd.multi_layout_for(0, 4, m, l, n)
    assert lay == m4ri_amd.LAYOUT_CYCLIC2
    dA, dB, dC = Dmat(m, l, lay).fill(3), Dmat(l, n, lay).fill(4), Dmat(m, n, lay)
    m4ri_amd.dmat_mul(dC, dA, dB)
    m4ri_amd.multi_sync()
    st = m4ri_amd.multi_stats()
    assert (st.variant, st.world, st.converted, st.levels, st.sub_products, st.chunks) == (m4ri_amd.VARIANT_STRASSEN, 4, 0, 2, 47, 1) and st.group >= 2
    assert hashlib.sha256(dC.download().masked().tobytes()).hexdigest() == _golden(m, l, n, (3, 4))
    for d in (dA, dB, dC):
        d.free()
    m4ri_amd.lib().m4ri_amd_release_workspace()


def test_a_matrix_of_an_older_device_list_is_refused_not_used():
    m4ri_amd.set_devices([0, 0])
    d = Dmat(256, 256, m4ri_amd.LAYOUT_ROWS).fill(1)
    m4ri_amd.set_devices([0, 0, 0])
    e = Dmat(256, 256, m4ri_amd.LAYOUT_ROWS)         # rebuilds the ranks for the new list
    assert d.info().alive == 0 and e.info().alive == 1
    assert m4ri_amd.lib().m4ri_amd_dmat_mul(e.h, d.h, d.h, 0, 0, 0) != 0
    d.free()
    e.free()


@pytest.mark.parametrize("lay", [m4ri_amd.LAYOUT_ROWS, m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.LAYOUT_CYCLIC2])
def test_two_products_in_flight_on_two_lanes(oracle, lay):
    """m4ri_amd_dmat_mul_lane: independent products issued alternately on lanes 0 and 1 (own streams, events, arenas per lane, the
    inputs shared) -- every result bit for bit the oracle's, whatever runs under what; operations that are not products (upload,
    download) order themselves behind both lanes."""
    m4ri_amd.set_devices([0] * 4)
    m, l, n = 2048, 1536, 2304
    mats = [(Mzd.random(m, l, 30 + k), Mzd.random(l, n, 40 + k)) for k in range(4)]
    want = [oracle.mul(None, a, b, 0) for a, b in mats]
    dA = [Dmat(m, l, lay).upload(a) for a, _ in mats]
    dB = [Dmat(l, n, lay).upload(b) for _, b in mats]
    dC = [Dmat(m, n, lay) for _ in mats]
    for rep in range(3):                      # the same buffers again: events and arenas of both lanes are reused
        for k in range(4):
            m4ri_amd.dmat_mul(dC[k], dA[k], dB[k], lane=k & 1)
        for k in range(4):
            assert dC[k].download().equal(want[k]), (lay, rep, k)
    # a new operand uploaded while products are in flight on both lanes, then used on lane 1 first
    A2 = Mzd.random(m, l, 77)
    m4ri_amd.dmat_mul(dC[0], dA[0], dB[0], lane=0)
    m4ri_amd.dmat_mul(dC[1], dA[1], dB[1], lane=1)
    dA[2].upload(A2)
    m4ri_amd.dmat_mul(dC[2], dA[2], dB[2], lane=1)
    m4ri_amd.dmat_mul(dC[3], dA[2], dB[3], lane=0)
    assert dC[2].download().equal(oracle.mul(None, A2, mats[2][1], 0)) and dC[3].download().equal(oracle.mul(None, A2, mats[3][1], 0))
    assert dC[0].download().equal(want[0]) and dC[1].download().equal(want[1])
    # accumulate onto the result of the other lane: C2 += A*B after C2 = ... on lane 1 (same C: the caller orders them with a sync)
    m4ri_amd.multi_sync()
    m4ri_amd.dmat_mul(dC[2], dA[0], dB[0], add=True, lane=0)
    assert dC[2].download().equal(oracle.addmul(oracle.mul(None, A2, mats[2][1], 0), mats[0][0], mats[0][1], 0))
    for d in dA + dB + dC:
        d.free()


def test_lanes_at_a_baseline_shape_match_the_single_gpu_product():
    """Two 32768^3 products in flight on 8 ranks (Strassen schedule, two row chunks each) against the one-GPU engine."""
    m4ri_amd.set_devices([0] * 8)
    n, w = 32768, 32768 // 64
    lay = m4ri_amd.LAYOUT_CYCLIC1
    dA, dB = Dmat(n, n, lay).fill(3), Dmat(n, n, lay).fill(4)
    dC = [Dmat(n, n, lay), Dmat(n, n, lay)]
    for k in range(6):
        m4ri_amd.dmat_mul(dC[k & 1], dA, dB, lane=k & 1)
    A = torch.empty((n, w), dtype=torch.int64, device="cuda")
    B, C = torch.empty_like(A), torch.empty_like(A)
    m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 3)
    m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 4)
    m4ri_amd.multi_sync()
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, False, 0)
    torch.cuda.synchronize()
    want = C.cpu().numpy()
    for k in range(2):
        assert np.array_equal(dC[k].download().valid_words().view(np.int64), want), k
    for d in [dA, dB] + dC:
        d.free()
    m4ri_amd.lib().m4ri_amd_release_workspace()


@pytest.mark.parametrize("spec,staged", [("all", 12), ("0-2,1-3", 4)])
def test_pairs_without_peer_access_copy_through_the_host_and_say_so(oracle, monkeypatch, spec, staged):
    """The injected "no peer access" flag (M4RI_AMD_NO_PEER) sends those pairs' copies through pinned host memory: still bit-exact in
    every schedule, and the stats count the pairs.  M4RI_AMD_SELFTEST_PAIRS=1 makes the first-contact self-test (one small copy behind
    one cross-rank event wait per ordered pair) run between ranks of ONE device too: it must pass before any product is scheduled."""
    monkeypatch.setenv("M4RI_AMD_NO_PEER", spec)
    monkeypatch.setenv("M4RI_AMD_SELFTEST_PAIRS", "1")
    m4ri_amd.set_devices([0] * 3)            # another device list first: the ranks are rebuilt under the new environment
    Dmat(64, 64, m4ri_amd.LAYOUT_ROWS).free()
    m4ri_amd.set_devices([0] * 4)
    try:
        for lay, variant in ((m4ri_amd.LAYOUT_ROWS, m4ri_amd.VARIANT_SLABS), (m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.VARIANT_STRASSEN),
                             (m4ri_amd.LAYOUT_CYCLIC2, m4ri_amd.VARIANT_STRASSEN)):
            m, l, n = 3000, 2048, 2500
            A, B = Mzd.random(m, l, 61), Mzd.random(l, n, 62)
            dA, dB, dC = Dmat(m, l, lay).upload(A), Dmat(l, n, lay).upload(B), Dmat(m, n, lay)
            m4ri_amd.dmat_mul(dC, dA, dB)
            st = m4ri_amd.multi_stats()
            assert (st.variant, st.pairs_staged) == (variant, staged), (st.variant, st.pairs_staged)
            want = oracle.mul(None, A, B, 0)
            assert dC.download().equal(want), (spec, lay)
            e = Dmat(m, n, m4ri_amd.LAYOUT_REPLICATED).convert_from(dC)     # the all-gather through the same copies
            assert e.download().equal(want)
            m4ri_amd.dmat_mul(dC, dA, dB, lane=1)
            assert dC.download().equal(want), (spec, lay, "lane 1")
            for d in (dA, dB, dC, e):
                d.free()
        probe = m4ri_amd.multi_link_probe(8 << 20)
        assert sum(1 - x for row in probe["peer_access"] for x in row) == staged
    finally:
        monkeypatch.delenv("M4RI_AMD_NO_PEER")
        monkeypatch.delenv("M4RI_AMD_SELFTEST_PAIRS")
        m4ri_amd.set_devices([0] * 3)
        Dmat(64, 64, m4ri_amd.LAYOUT_ROWS).free()


def test_link_probe_reports_every_ordered_pair():
    m4ri_amd.set_devices([0] * 4)
    p = m4ri_amd.multi_link_probe(32 << 20)
    assert p["pairs"] == 12 and p["ranks_share_devices"] is True and p["gbs_per_direction_min"] > 1.0 and p["all_at_once_gbs"] > 1.0
    assert all(p["pair_gbs"][i][i] == 0 for i in range(4)) and all(all(row) for row in p["peer_access"])
