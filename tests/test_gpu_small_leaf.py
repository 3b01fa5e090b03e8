"""The one-launch leaf for small products (m4ri_amd/csrc/m4rm_small.hip, "generation 5"; engine.hip: small_leaf_wanted) through the
device-pointer entry points, against the oracle's gf2o_mul / gf2o_addmul (reference brilliantrussian.c:1032-1190 semantics: only C's bits
are observable) and against the generation-4 path on the same operands."""
import numpy as np
import pytest
import torch

import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)
    torch.cuda.set_device(0)


def _views(hA, hB, hC, pad, dirty):
    """Operands as views into larger device buffers: row strides `pad` words longer than the rows, the views starting `pad` words in;
    dirty: the bits of A beyond its last column and of B beyond its last column hold junk (what a window of a wider parent carries)."""
    m, l, n = hA.nrows, hA.ncols, hB.ncols
    wl, wn = (l + 63) // 64, (n + 63) // 64
    out = []
    for h, rows, w, ncols in ((hA, m, wl, l), (hB, l, wn, n), (hC, m, wn, n)):
        words = h.valid_words().copy()
        if dirty and h is not hC and ncols % 64:
            junk = np.random.default_rng(ncols).integers(0, 2 ** 63, size=rows, dtype=np.int64).astype(np.uint64)
            words[:, w - 1] |= junk << np.uint64(ncols % 64)
        t = torch.full((rows + 2, w + 2 * pad), -1, dtype=torch.int64, device="cuda")
        t[1:rows + 1, pad:pad + w] = torch.from_numpy(words.view(np.int64).copy()).cuda()
        out.append((t, t.data_ptr() + 8 * ((w + 2 * pad) + pad), w + 2 * pad))
    return out


def _valid(t, rows, ncols, pad):
    w = (ncols + 63) // 64
    got = t[1:rows + 1, pad:pad + w].cpu().numpy().view(np.uint64).copy()
    if ncols % 64:
        got[:, w - 1] &= np.uint64((1 << (ncols % 64)) - 1)
    return got


SHAPES = [(1, 1, 1), (1, 64, 1), (2, 65, 3), (64, 64, 64), (255, 63, 65), (256, 64, 512), (257, 129, 513), (300, 700, 100), (100, 1000, 100),
          (1000, 64, 1000), (511, 1025, 1023), (1100, 1290, 1411), (2048, 2048, 2048), (33, 4096, 70), (4096, 16, 64), (64, 64, 4160), (3000, 130, 40)]


@pytest.mark.parametrize("m,l,n", SHAPES)
@pytest.mark.parametrize("add", [False, True])
def test_small_products_match_the_oracle(oracle, m, l, n, add):
    hA, hB, hC = Mzd.random(m, l, 7 + m), Mzd.random(l, n, 8 + l), Mzd.random(m, n, 9 + n)
    want = oracle.addmul(hC.copy(), hA, hB, 0) if add else oracle.mul(None, hA, hB, 0)
    for pad, dirty in ((0, False), (3, True)):
        (tA, pA, sA), (tB, pB, sB), (tC, pC, sC) = _views(hA, hB, hC, pad, dirty)
        before = tC.clone()
        m4ri_amd.mul_dev(pC, sC, pA, sA, pB, sB, m, l, n, add=add)
        torch.cuda.synchronize()
        st = m4ri_amd.get_stats()
        assert st.levels == 0 and st.leaf_gen == 5, (st.levels, st.leaf_gen)
        assert np.array_equal(_valid(tC, m, n, pad), want.valid_words()), (pad, dirty)
        # nothing outside the rows' own words is touched: the frame of -1 around the view is intact
        wn = (n + 63) // 64
        mask = torch.ones_like(tC, dtype=torch.bool)
        mask[1:m + 1, pad:pad + wn] = False
        assert torch.equal(tC[mask], before[mask])


@pytest.mark.parametrize("m,l,n", [(512, 512, 512), (300, 5000, 200), (1500, 1500, 1500), (70, 70, 9000)])
@pytest.mark.parametrize("add", [False, True])
def test_small_leaf_and_generation_4_give_the_same_words(m, l, n, add):
    """Whole words, excess columns included: m4rm_dev with an explicit inner split keeps the older kernels (the engine never hands a
    caller's split to the small leaf), ksplit = 0 lets the engine choose -- the small leaf at these sizes."""
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
    C0 = torch.empty((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 21)
    m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 22)
    m4ri_amd.fill_dev(C0.data_ptr(), wn, m, n, 23)
    out = []
    for ksplit, gen in ((0, 5), (1, None)):
        C = C0.clone()
        m4ri_amd.m4rm_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add=add, ksplit=ksplit)
        torch.cuda.synchronize()
        if gen is not None:
            assert m4ri_amd.get_stats().leaf_gen == gen
        else:
            assert m4ri_amd.get_stats().leaf_gen != 5
        out.append(C)
    assert torch.equal(out[0], out[1])


def test_the_rule_is_by_work():
    """Up to 2^34 bit operations (batch included) a direct product takes the small leaf; above, and whenever the fused passes have
    written the packed A, generation 4 (or generation 1 for thin shapes)."""
    def gen_of(m, l, n):
        wl, wn = (l + 63) // 64, (n + 63) // 64
        A = torch.zeros((m, wl), dtype=torch.int64, device="cuda")
        B = torch.zeros((l, wn), dtype=torch.int64, device="cuda")
        C = torch.zeros((m, wn), dtype=torch.int64, device="cuda")
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n)
        torch.cuda.synchronize()
        st = m4ri_amd.get_stats()
        return st.levels, st.leaf_gen
    assert gen_of(2048, 2048, 2048) == (0, 5)
    assert gen_of(200, 8192, 8192) == (0, 5)
    assert gen_of(4096, 4096, 4096) == (0, 4)
    assert gen_of(464, 16384, 16421)[1] == 1
