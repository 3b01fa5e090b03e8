"""Parity proper (needs an MI355X): the HIP path, called through the C ABI of libm4ri_amd.so, against
the CPU oracle on the same seeded inputs, against the committed golden fixtures (generated from the
real reference), and -- at BASELINE.json's full sizes -- against reference fingerprints and
size-independent identities.  Bit-exact everywhere: this is integer work.

Shapes are the reference's own (tests/test_multiplication.c:251-322, tests/test_smallops.c:115-121)
plus empty / ragged / word-boundary edges.
"""
import os

import numpy as np
import pytest

import m4ri_amd
import shapes
from m4ri_amd.mzd import Mzd
from test_golden_oracle import GOLD, load_kats, run_kat

pytestmark = pytest.mark.gpu


def sha_matches(M, op, m, l, n, seed_a, cutoff=0):
    """SHA-256 of the valid words of host matrix M (row-major, excess masked) against the reference's for the
    same inputs (tests/golden/sha256.json, make_golden.py --sha): beside the 64-bit FNV fingerprints."""
    import hashlib
    import json
    p = golden_file("sha256.json")
    for e in json.load(open(p)):
        if (e["op"], e["m"], e["l"], e["n"], e["seed_a"], e["cutoff"]) == (op, m, l, n, seed_a, cutoff):
            return hashlib.sha256(M.masked().tobytes()).hexdigest() == e["sha256"]
    raise AssertionError(f"tests/golden/sha256.json holds no entry for {op} {m}x{l}x{n} seed {seed_a} cutoff {cutoff}: "
                         "a lost fixture must fail the test, not soften it")


def golden_file(name):
    """A committed fixture of tests/golden: its absence is a test failure (the parity claim rests on it), never a skip."""
    p = os.path.join(GOLD, name)
    assert os.path.exists(p), f"committed fixture tests/golden/{name} is missing"
    return p


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    L = m4ri_amd.lib()
    assert L.m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)


def _pair(m, l, n, tag):
    return Mzd.random(m, l, shapes.seed_of(tag, m, l, n, 1)), Mzd.random(l, n, shapes.seed_of(tag, m, l, n, 2))


# ---- the reference's own differential shapes ------------------------------------------------------
@pytest.mark.parametrize("m,l,n,k,cutoff", shapes.MUL + shapes.EDGE)
def test_mul(oracle, m, l, n, k, cutoff):
    A, B = _pair(m, l, n, 31)
    want = oracle.mul(None, A, B, cutoff)
    assert m4ri_amd.mzd_mul(None, A, B, cutoff).equal(want)            # C == NULL: allocated result
    assert m4ri_amd.mzd_mul(Mzd.random(m, n, 5), A, B, cutoff).equal(want)  # dirty C is overwritten
    assert m4ri_amd.mzd_mul_m4rm(None, A, B, k).equal(want)            # leaf only
    assert m4ri_amd.mzd_mul_mp(None, A, B, cutoff).equal(want)
    assert m4ri_amd.mzd_mul_mp(Mzd.random(m, n, 7), A, B, cutoff).equal(want)  # dirty C is overwritten (the reference's adds its strips: test_oracle_vs_reference)
    if m and n:
        assert m4ri_amd._mzd_mul_even(Mzd.init(m, n), A, B, max(64, cutoff)).equal(want)
        assert m4ri_amd._mzd_mul_m4rm(Mzd.random(m, n, 6), A, B, k, 1).equal(want)


@pytest.mark.parametrize("m,l,n,k,cutoff", shapes.ADDMUL)
def test_addmul(oracle, m, l, n, k, cutoff):
    A, B = _pair(m, l, n, 32)
    C0 = Mzd.random(m, n, shapes.seed_of(32, m, l, n, 3))
    want = oracle.addmul(C0.copy(), A, B, cutoff)
    assert m4ri_amd.mzd_addmul(C0.copy(), A, B, cutoff).equal(want)
    assert m4ri_amd.mzd_addmul_m4rm(C0.copy(), A, B, k).equal(want)
    assert m4ri_amd._mzd_addmul(C0.copy(), A, B, max(64, cutoff)).equal(want)
    assert m4ri_amd._mzd_addmul_even(C0.copy(), A, B, max(64, cutoff)).equal(want)
    assert m4ri_amd._mzd_mul_m4rm(C0.copy(), A, B, k, 0).equal(want)
    assert m4ri_amd.mzd_addmul_mp(C0.copy(), A, B, cutoff).equal(want)


@pytest.mark.parametrize("n,k,cutoff", shapes.SQR)
def test_sqr_aliasing(oracle, n, k, cutoff):
    A = Mzd.random(n, n, shapes.seed_of(33, n))
    want = oracle.mul(None, A, A, cutoff)
    assert m4ri_amd.mzd_mul(None, A, A, cutoff).equal(want)  # A == B, same pointer (strassen.c:363)
    L = m4ri_amd.lib()
    C = Mzd.init(n, n)
    L._mzd_sqr_even(C.ptr, A.ptr, max(64, cutoff))
    assert C.equal(want)


@pytest.mark.parametrize("n,k,cutoff", shapes.ADDSQR)
def test_addsqr_aliasing(oracle, n, k, cutoff):
    A = Mzd.random(n, n, shapes.seed_of(34, n))
    C0 = Mzd.random(n, n, shapes.seed_of(34, n, 3))
    want = oracle.addmul(C0.copy(), A, A, cutoff)
    assert m4ri_amd.mzd_addmul(C0.copy(), A, A, cutoff).equal(want)
    C = C0.copy()
    m4ri_amd.lib()._mzd_addsqr_even(C.ptr, A.ptr, max(64, cutoff))
    assert C.equal(want)


@pytest.mark.parametrize("m,l,n,cutoff", [(111, 111, 111, 64), (248, 92, 1024, 64), (127, 300, 86, 64), (300, 127, 127, 64), (86, 86, 86, 64),
                                          (200, 200, 200, 128), (255, 171, 255, 128)])
def test_hints_the_reference_recursion_cannot_take(oracle, m, l, n, cutoff):
    """A cutoff of 64 (128) with a dimension in 86 .. 127 (171 .. 255) leaves the reference's own recursion with empty quadrants:
    its _mzd_addmul_even and _mzd_sqr_even (strassen.c:396-420, :210-240, no guard for them) abort in mzd_copy or read out of bounds
    (found by tests/soak_mul.py).  A hint changes no bit, so the result is the one of cutoff 0 -- which is what the library gives."""
    A, B = _pair(m, l, n, 35)
    C0 = Mzd.random(m, n, shapes.seed_of(35, m, l, n, 3))
    want, want_add = oracle.mul(None, A, B, 0), oracle.addmul(C0.copy(), A, B, 0)
    assert m4ri_amd.mzd_mul(None, A, B, cutoff).equal(want)
    assert m4ri_amd.mzd_addmul(C0.copy(), A, B, cutoff).equal(want_add)
    assert m4ri_amd._mzd_addmul_even(C0.copy(), A, B, cutoff).equal(want_add)
    if m == l == n:
        sq, sq_add = oracle.mul(None, A, A, 0), oracle.addmul(C0.copy(), A, A, 0)
        assert m4ri_amd.mzd_mul(None, A, A, cutoff).equal(sq)
        C = C0.copy()
        m4ri_amd.lib()._mzd_addsqr_even(C.ptr, A.ptr, cutoff)
        assert C.equal(sq_add)


@pytest.mark.parametrize("m,l,n", shapes.EMPTY_INNER)
def test_empty_inner_dimension(m, l, n):
    A, B = Mzd.init(m, l), Mzd.init(l, n)
    C = Mzd.random(m, n, 9)
    keep = C.copy()
    assert m4ri_amd.mzd_addmul(C, A, B, 0).equal(keep)      # C += 0
    assert m4ri_amd.mzd_mul(C, A, B, 0).equal(Mzd.init(m, n))  # C = 0


# ---- windows with non-zero excess inside a pattern-filled parent (test_smallops.c) --------------
@pytest.mark.parametrize("M,N,m,n", shapes.SMALLOPS)
@pytest.mark.parametrize("cutoff", [0, 64])
def test_windows_preserve_parent(oracle, M, N, m, n, cutoff):
    pat = np.uint64(shapes.SMALLOPS_PATTERN)

    def parent():
        P = Mzd.init(M, N)
        P.rows()[:, :] = pat
        return P

    PA, PB, PCg, PCo = parent(), parent(), parent(), parent()
    k = min(m, n)
    a, b = PA.window(0, 0, m, k), PB.window(0, 0, k, n)
    a.fill_splitmix(shapes.seed_of(35, M, N, 1))
    b.fill_splitmix(shapes.seed_of(35, M, N, 2))
    cg, co = PCg.window(0, 0, m, n), PCo.window(0, 0, m, n)
    cg.fill_splitmix(78)
    co.fill_splitmix(78)
    m4ri_amd.mzd_mul(cg, a, b, cutoff)
    oracle.mul(co, a, b, cutoff)
    assert np.array_equal(PCg.buf, PCo.buf)  # every word of the parent, pattern and excess bits included
    m4ri_amd.mzd_addmul(cg, a, b, cutoff)
    oracle.addmul(co, a, b, cutoff)
    assert np.array_equal(PCg.buf, PCo.buf)
    m4ri_amd.mzd_addmul_m4rm(cg, a, b, 0)
    oracle.mul_m4rm(co, a, b, 0, 0)
    assert np.array_equal(PCg.buf, PCo.buf)
    # a window that does not start at (0,0): rows and word-aligned columns offset
    if M >= 2 * m and N >= 64 + n:
        cg2, co2 = PCg.window(M - m, 64, M, 64 + n), PCo.window(M - m, 64, M, 64 + n)
        m4ri_amd.mzd_mul(cg2, a, b, cutoff)
        oracle.mul(co2, a, b, cutoff)
        assert np.array_equal(PCg.buf, PCo.buf)
    assert np.all(PA.rows()[m:, :] == pat) and np.all(PB.rows()[k:, :] == pat)  # operands only read


def test_smallops_identity():
    """(A+B)^2 == A^2 + BA + AB + B^2 through mzd_mul / mzd_addmul (test_smallops.c:70-85)."""
    for n in (64, 513, 1024):
        A, B = Mzd.random(n, n, 41), Mzd.random(n, n, 42)
        S = Mzd.init(n, n)
        S.valid_words()[:, :] = A.valid_words() ^ B.valid_words()
        D = m4ri_amd.mzd_mul(None, S, S, 0)
        C = m4ri_amd.mzd_mul(None, A, A, 0)
        m4ri_amd.mzd_addmul(C, B, A, 0)
        m4ri_amd.mzd_addmul(C, A, B, 0)
        m4ri_amd.mzd_addmul(C, B, B, 0)
        assert C.equal(D)


# ---- golden fixtures from the real reference -----------------------------------------------------
@pytest.mark.parametrize("case", list(load_kats()), ids=lambda c: f"{c[1]}-{c[2]}x{c[3]}x{c[4]}-{c[5]}")
def test_golden_kat(case):
    i, op, m, l, n, par, seeds, Aw, Bw, Cw = case
    got = run_kat(m4ri_amd.mzd_mul, m4ri_amd.mzd_addmul, lambda C, A, B, k: m4ri_amd.mzd_mul_m4rm(C, A, B, k),
                  op, m, l, n, par, seeds, Aw, Bw)
    assert np.array_equal(got.masked(), Cw)


def test_golden_window_parents():
    z = np.load(os.path.join(GOLD, "kat_small.npz"))
    for (M, N, m, n) in shapes.SMALLOPS[:2]:
        pat = np.uint64(shapes.SMALLOPS_PATTERN)
        PA, PB, PC = Mzd.init(M, N), Mzd.init(M, N), Mzd.init(M, N)
        for P in (PA, PB, PC):
            P.rows()[:, :] = pat
        k = min(m, n)
        a, b, c = PA.window(0, 0, m, k), PB.window(0, 0, k, n), PC.window(0, 0, m, n)
        a.fill_splitmix(shapes.seed_of(22, M, N, 1))
        b.fill_splitmix(shapes.seed_of(22, M, N, 2))
        c.fill_splitmix(shapes.seed_of(22, M, N, 3))
        m4ri_amd.mzd_addmul(c, a, b, 0)
        assert np.array_equal(PC.buf, z[f"win{M}_{N}_{m}_{n}_parentC"])


def test_golden_fingerprints_host_api(oracle):
    """Large products of the reference, inputs regenerated from seeds (incl. the 4096x3528x4096
    regression shape named in strassen.c:127-134, 8192^3, a ragged 3000x5000x7001 and 16384^3)."""
    z = np.load(os.path.join(GOLD, "fingerprints.npz"))
    for op, (m, l, n, par), (sa, sb, sc), fp in zip(z["ops"], z["meta"], z["seeds"], z["fp"]):
        m, l, n, par = int(m), int(l), int(n), int(par)
        A, B = Mzd.random(m, l, int(sa)), Mzd.random(l, n, int(sb))
        if op == "mul":
            C = m4ri_amd.mzd_mul(None, A, B, par)
        elif op == "m4rm":
            C = m4ri_amd.mzd_mul_m4rm(None, A, B, par)
        else:
            C = m4ri_amd.mzd_addmul(Mzd.random(m, n, int(sc)), A, B, par)
        assert oracle.fingerprint(C) == int(fp), (op, m, l, n, par)
        # cutoff / k are hints: other values give the same bits
        if op == "mul" and m * l * n <= 4096 ** 3:
            for other in (64, 512, 1 << 20):
                assert oracle.fingerprint(m4ri_amd.mzd_mul(None, A, B, other)) == int(fp)


# ---- device-resident API at BASELINE sizes ----------------------------------------------------
torch = pytest.importorskip("torch")


def dev_random(rows, cols, seed):
    w = (cols + 63) // 64
    t = torch.empty((rows, w), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(t.data_ptr(), w, rows, cols, seed)
    return t


def to_host(t, rows, cols):
    M = Mzd(rows, cols, rowstride=t.shape[1])
    torch.cuda.synchronize()
    M.rows()[:, :] = t.cpu().numpy().view(np.uint64)
    return M


def test_device_fill_matches_host_fill():
    for (r, c) in [(3, 1), (5, 65), (100, 4096), (7, 200)]:
        t = dev_random(r, c, 1234)
        assert np.array_equal(to_host(t, r, c).masked(), Mzd.random(r, c, 1234).masked())


def test_config2_leaf_16384_vs_reference_fingerprint(oracle):
    """BASELINE.json configs[1]: 16384^3 through the M4RM leaf only, all 32 MiB of C compared with the
    reference's product by fingerprint (golden), seeds as in the fixture."""
    z = np.load(os.path.join(GOLD, "fingerprints.npz"))
    i = [k for k, mt in enumerate(z["meta"]) if tuple(int(x) for x in mt[:3]) == (16384, 16384, 16384)]
    assert i, "tests/golden/fingerprints.npz holds no 16384^3 entry (BASELINE.json configs[1])"
    sa, sb, _ = (int(x) for x in z["seeds"][i[0]])
    n = 16384
    A, B = dev_random(n, n, sa), dev_random(n, n, sb)
    C = torch.empty((n, n // 64), dtype=torch.int64, device="cuda")
    m4ri_amd.m4rm_dev(C.data_ptr(), n // 64, A.data_ptr(), n // 64, B.data_ptr(), n // 64, n, n, n)
    assert oracle.fingerprint(to_host(C, n, n)) == int(z["fp"][i[0]])
    # and the Strassen engine gives the same bits
    C2 = torch.empty_like(C)
    m4ri_amd.mul_dev(C2.data_ptr(), n // 64, A.data_ptr(), n // 64, B.data_ptr(), n // 64, n, n, n)
    assert torch.equal(C, C2)
    st = m4ri_amd.get_stats()
    assert st.levels == m4ri_amd.plan_levels(n, n, n, 0) == 2 and st.leaf_products == _scheme_leaves(2)   # the engine's own depth: leaves of 4096^3


def freivalds(oracle, A, B, C, m, l, n, seed):
    """C == A*B  <=>  C*x == A*(B*x) for a random n x 64 block x (error probability 2^-64), with the
    thin products done on the CPU by the oracle."""
    x = Mzd.random(n, 64, seed)
    Bx = oracle.mul_m4rm(Mzd.init(l, 64), B, x, 8, 1)
    ABx = oracle.mul_m4rm(Mzd.init(m, 64), A, Bx, 8, 1)
    Cx = oracle.mul_m4rm(Mzd.init(m, 64), C, x, 8, 1)
    return ABx.equal(Cx)


def test_config3_65536_strassen_properties(oracle):
    """BASELINE.json configs[2]: 65536^3 mzd_mul on one GPU (seeds 3, 4).  Checked by (i) the
    reference's fingerprint when the XL fixture exists, (ii) Freivalds' identity on the CPU,
    (iii) linearity C(A1+A2, B) == C(A1, B) + C(A2, B), (iv) agreement of two different schedules."""
    n = 65536
    w = n // 64
    A, B = dev_random(n, n, 3), dev_random(n, n, 4)
    C = torch.empty((n, w), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
    st = m4ri_amd.get_stats()
    assert st.levels == 4 and st.leaf_products == _scheme_leaves(4) and (st.leaf_m, st.leaf_l, st.leaf_n) == (4096, 4096, 4096)
    hC = to_host(C, n, n)
    z = np.load(golden_file("fingerprints_xl.npz"))
    i = [k for k, mt in enumerate(z["meta"]) if tuple(int(x) for x in mt[:3]) == (n, n, n)]
    assert i, "fingerprints_xl.npz holds no 65536^3 entry"
    assert oracle.fingerprint(hC) == int(z["fp"][i[0]])
    assert sha_matches(hC, "mul", n, n, n, 3)
    assert freivalds(oracle, to_host(A, n, n), to_host(B, n, n), hC, n, n, n, 77)
    # different schedule (2 levels, 16384^3 leaves), same bits
    C2 = torch.empty_like(C)
    m4ri_amd.mul_dev(C2.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, cutoff=16384)
    assert m4ri_amd.get_stats().levels == 2
    assert torch.equal(C, C2)
    # linearity, with the accumulate path: C ^= A2*B must equal (A ^ A2)*B
    A2 = dev_random(n, n, 33)
    m4ri_amd.mul_dev(C.data_ptr(), w, A2.data_ptr(), w, B.data_ptr(), w, n, n, n, add=True)
    m4ri_amd.xor_dev(A2.data_ptr(), w, A2.data_ptr(), w, A.data_ptr(), w, n, n)
    m4ri_amd.mul_dev(C2.data_ptr(), w, A2.data_ptr(), w, B.data_ptr(), w, n, n, n)
    assert torch.equal(C, C2)


def test_config5_rectangular_131072(oracle):
    """BASELINE.json configs[4] shape on one GPU: 131072 x 8192 x 131072 (C = 2 GiB), seeds 5, 6."""
    m, l, n = 131072, 8192, 131072
    A, B = dev_random(m, l, 5), dev_random(l, n, 6)
    C = torch.empty((m, n // 64), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), n // 64, A.data_ptr(), l // 64, B.data_ptr(), n // 64, m, l, n)
    hC = to_host(C, m, n)
    z = np.load(golden_file("fingerprints_xl.npz"))
    i = [k for k, mt in enumerate(z["meta"]) if tuple(int(x) for x in mt[:3]) == (m, l, n)]
    assert i, "fingerprints_xl.npz holds no 131072 x 8192 x 131072 entry"
    assert oracle.fingerprint(hC) == int(z["fp"][i[0]])
    assert sha_matches(hC, "mul", m, l, n, 5)
    assert freivalds(oracle, to_host(A, m, l), to_host(B, l, n), hC, m, l, n, 78)


def test_device_views_and_ragged_strassen(oracle):
    """Strided device views + dimensions that leave all three remainder strips (strassen.c:170-204)."""
    m, l, n = 2 * 8192 + 37, 2 * 8192 + 64 + 5, 2 * 8192 + 128 + 11
    hA, hB = Mzd.random(m, l, 51), Mzd.random(l, n, 52)
    wa, wn = hA.rowstride, hB.rowstride
    A = torch.from_numpy(hA.rows().view(np.int64).copy()).cuda()
    B = torch.from_numpy(hB.rows().view(np.int64).copy()).cuda()
    C = torch.zeros((m, wn + 3), dtype=torch.int64, device="cuda")  # padded stride
    # (cutoff given: the engine's own depth model keeps this shape unsplit -- its strips cost more than one level saves)
    m4ri_amd.mul_dev(C.data_ptr(), wn + 3, A.data_ptr(), wa, B.data_ptr(), wn, m, l, n, cutoff=8192)
    assert m4ri_amd.get_stats().levels == m4ri_amd.plan_levels(m, l, n, 8192) >= 1
    got = to_host(C, m, n)
    want = oracle.mul(None, hA, hB, 8192)
    assert got.equal(want)
    assert np.all(got.rows()[:, got.width:] == 0)  # padding words never written


@pytest.mark.parametrize("m,l,n,cutoff,a_shift,leaf_gen", [
    (8192, 8192, 8192, 2048, 0, 4),    # two fused levels, 2048-row leaves (half-filled tiles): pass writes the rotated packed A
    (16384, 4096, 4096, 1024, 0, 4),   # 4096-row leaves, short inner dimension
    (8192, 5120, 8192, 1024, 0, 4),    # 20-word leaf rows do not tile the fused pass: down2 + separate pack
    (8192, 8192, 8192, 2048, 1, 4),    # A view starting one word into its rows (8-byte aligned, odd stride)
])
def test_fused_down_pack_paths(oracle, m, l, n, cutoff, a_shift, leaf_gen):
    """The last A-side pass of the breadth-first schedule writes the leaf's packed operand directly
    (engine.hip bfs_product); shapes and views that cannot take it fall back to pass + pack."""
    hA, hB = Mzd.random(m, l, 61), Mzd.random(l, n, 62)
    wa, wn = hA.rowstride, hB.rowstride
    At = torch.zeros((m, wa + a_shift), dtype=torch.int64, device="cuda")
    At[:, a_shift:] = torch.from_numpy(hA.rows().view(np.int64).copy()).cuda()
    B = torch.from_numpy(hB.rows().view(np.int64).copy()).cuda()
    C = torch.empty((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), wn, At.data_ptr() + 8 * a_shift, wa + a_shift, B.data_ptr(), wn, m, l, n, cutoff=cutoff)
    st = m4ri_amd.get_stats()
    assert st.levels == 2 and st.leaf_gen == leaf_gen
    assert to_host(C, m, n).equal(oracle.mul(None, hA, hB, cutoff))


@pytest.mark.parametrize("m,l,n,cutoff,leaf_gen,add", [
    (16384, 16384, 16384, 2048, 4, False),  # 343 leaves of 2048^3: three-level pass writes the packed A of half-filled tiles
    (32768, 8192, 8192, 1024, 4, False),    # 4096-row leaves
    (8192, 10240, 8192, 1024, 4, False),    # 20-word leaf rows of 1024-row leaves: plain three-level passes + separate pack
    (4096, 4096, 4096, 512, 4, True),       # accumulate through the three-level up pass
])
def test_three_level_fused_passes(oracle, m, l, n, cutoff, leaf_gen, add):
    """One fused pass per operand for the three deepest levels (aux_kernels.hip winograd_down3 / up3);
    fusing 1, 2 or 3 levels is a scheduling choice and must not change a bit."""
    hA, hB, hC = Mzd.random(m, l, 71), Mzd.random(l, n, 72), Mzd.random(m, n, 73)
    wa, wn = hA.rowstride, hB.rowstride
    A = torch.from_numpy(hA.rows().view(np.int64).copy()).cuda()
    B = torch.from_numpy(hB.rows().view(np.int64).copy()).cuda()
    C0 = torch.from_numpy(hC.rows().view(np.int64).copy()).cuda()
    want = oracle.addmul(hC.copy(), hA, hB, cutoff) if add else oracle.mul(None, hA, hB, cutoff)
    old = m4ri_amd.set_max_fuse(0)
    try:
        for fuse in (3, 2, 1):
            m4ri_amd.set_max_fuse(fuse)
            C = C0.clone()
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wa, B.data_ptr(), wn, m, l, n, add=add, cutoff=cutoff)
            st = m4ri_amd.get_stats()
            assert st.levels == 3 and st.leaf_gen == leaf_gen
            assert to_host(C, m, n).equal(want), f"max_fuse={fuse}"
    finally:
        m4ri_amd.set_max_fuse(old)


@pytest.mark.parametrize("m,l,n,cutoff,add,strided", [
    (8192, 8192, 8192, 512, False, False),     # 2401 leaves of 512^3: the four-level passes, packed A from the pass itself
    (4096, 5120, 4096, 256, True, False),      # accumulate: the atomic top level adds onto C; 5-word leaf rows: plain passes + separate pack
    (16384, 8192, 8192, 512, False, True),     # operands and result inside wider parents (strided ancestors), 1024-row leaves
    (16384 + 40, 8192 + 70, 8192 + 130, 512, True, False),   # remainder strips around the even block
    (4096, 4096, 32768, 256, True, False),     # 32-word leaf rows: the up pass whose products meet in LDS, accumulating
    (4096, 4096, 32768, 256, False, True),     # the same, plain, writing into a wider parent
    (4096, 16384, 4096, 256, False, False),    # 16-word leaf rows of A: the pack pass without the transpose (lane = row), one full group
    (2560, 16384, 2048, 128, True, True),      # the same with 5 row blocks of 32 (the group's other three workgroups idle), strided
    (4096, 32768, 4096, 256, False, False),    # two groups of 16 word columns per row block
])
def test_four_level_fused_passes(oracle, m, l, n, cutoff, add, strided):
    """Four levels in one pass each way (aux_kernels.hip winograd_down4 / down4_pack / up4: the three-level passes with the top level
    formed on the fly, the seven top-level products of a word meeting in LDS -- or, for leaf rows that are not a multiple of 32 words,
    by atomic XOR in HBM): a scheduling choice like the others -- fusing 4, 3 or 1 levels must not change a bit."""
    hA, hB, hC = Mzd.random(m, l, 81), Mzd.random(l, n, 82), Mzd.random(m, n, 83)
    pad = 6 if strided else 0          # words of a wider parent to the right of every operand
    wa, wn = hA.rowstride + pad, hB.rowstride + pad

    def dev(h, stride):
        t = torch.full((h.nrows, stride), -1, dtype=torch.int64, device="cuda")
        t[:, :h.rowstride] = torch.from_numpy(h.rows().view(np.int64).copy()).cuda()
        return t
    A, B, C0 = dev(hA, wa), dev(hB, wn), dev(hC, wn)
    want = oracle.addmul(hC.copy(), hA, hB, cutoff) if add else oracle.mul(None, hA, hB, cutoff)
    old = m4ri_amd.set_max_fuse(0)
    try:
        for fuse in (4, 3, 1):
            m4ri_amd.set_max_fuse(fuse)
            C = C0.clone()
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wa, B.data_ptr(), wn, m, l, n, add=add, cutoff=cutoff)
            st = m4ri_amd.get_stats()
            assert st.levels == 4
            got = Mzd(m, n)
            got.rows()[:, :] = C[:, :hC.rowstride].cpu().numpy().view(np.uint64)
            assert got.equal(want), f"max_fuse={fuse}"
            assert bool((C[:, hC.rowstride:] == -1).all()), f"max_fuse={fuse}: words of the parent outside C were written"
    finally:
        m4ri_amd.set_max_fuse(old)


def _scheme_rank():
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "m4ri_amd", "csrc", "scheme444.h")).read()
    return int(re.search(r"#define SCHEME444_R (\d+)", text).group(1))


def _scheme_leaves(levels):
    """Leaf products of `levels` Strassen levels as the engine runs them by default: the bottom (up to four) fused levels go through the
    4 x 4 x 4 scheme of rank R when the table beats Strassen applied twice (R < 49; scheme_passes.hip gf2_scheme444_ok, M4RI_AMD_SCHEME)."""
    R = _scheme_rank()
    sw = os.environ.get("M4RI_AMD_SCHEME")
    on = (R < 49) if sw is None else (int(sw) != 0)
    fused = min(levels, 4)
    if not on or fused < 2:
        return 7 ** levels
    return 7 ** (levels - fused) * {2: R, 3: 7 * R, 4: R * R}[fused]


@pytest.mark.parametrize("m,l,n,levels,add,strided", [
    (4096, 16384, 65536, 4, False, False),        # leaves 256 x 1024 x 4096: the smallest the scheme passes take; the scheme applied twice
    (4096, 32768, 65536, 4, True, True),          # accumulating into a wider parent; 32-word rows of A
    (4096 + 512, 16384, 65536, 4, False, False),  # leaves of 288 rows: nine row groups of 32, partly filled tiles
    (2048, 8192, 32768, 3, False, False),         # three levels: one Winograd level over the scheme
    (2048, 16384, 32768, 3, True, True),
    (1024, 4096, 16384, 2, False, False),         # two levels: the scheme once
    (1024, 8192, 16384, 2, True, True),
    (8192, 32768, 131072, 5, False, False)])      # five levels: a single Winograd pass over seven ancestors of the four-level scheme pass
def test_scheme_passes_match_the_winograd_passes(oracle, m, l, n, levels, add, strided):
    """The fused bottom levels through the rank-R scheme for the 4 x 4 x 4 block product (scheme_passes.hip, scheme444.h: R, 7 R or R^2 leaf
    products per ancestor instead of 7^2, 7^3, 7^4) against the same product through single-level Winograd passes (max_fuse 1: another
    kernel family, another number of leaves) -- identical bits -- and against the oracle through Freivalds' identity.  The stats say the
    scheme ran."""
    if os.environ.get("M4RI_AMD_SCHEME") != "1":
        # the library takes the scheme passes by itself only when the table beats Strassen applied twice; the test reaches them whatever the
        # table is by running itself in a child process with the switch set (read once per process)
        _in_child_with_scheme("1", "test_scheme_passes_match_the_winograd_passes", m, l, n, levels, add, strided)
        return
    _fused_against_single_level(oracle, m, l, n, levels, add, strided)


@pytest.mark.parametrize("m,l,n,levels,add,strided", [
    (4096, 16384, 65536, 4, False, False), (4096, 32768, 65536, 4, True, True), (2048, 16384, 32768, 3, True, True), (1024, 4096, 16384, 2, False, False)])
def test_winograd_passes_on_the_scheme_shapes_with_the_scheme_switched_off(oracle, m, l, n, levels, add, strided):
    """M4RI_AMD_SCHEME=0 gives the fused Winograd passes (aux_kernels.hip) back on the leaf shapes the scheme passes take by default since
    the table has rank 47: 7^levels leaf products, the same bits."""
    if os.environ.get("M4RI_AMD_SCHEME") != "0":
        _in_child_with_scheme("0", "test_winograd_passes_on_the_scheme_shapes_with_the_scheme_switched_off", m, l, n, levels, add, strided)
        return
    _fused_against_single_level(oracle, m, l, n, levels, add, strided)


def _in_child_with_scheme(value, name, m, l, n, levels, add, strided):
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        f"{name} and {m}-{l}-{n}-{levels}-{add}-{strided}"],
                       capture_output=True, text=True, env=dict(os.environ, M4RI_AMD_SCHEME=value), timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def _fused_against_single_level(oracle, m, l, n, levels, add, strided):
    hA, hB, hC = Mzd.random(m, l, 91), Mzd.random(l, n, 92), Mzd.random(m, n, 93)
    pad = 4 if strided else 0
    wa, wn = hA.rowstride + pad, hB.rowstride + pad

    def dev(h, stride):
        t = torch.full((h.nrows, stride), -1, dtype=torch.int64, device="cuda")
        t[:, :h.rowstride] = torch.from_numpy(h.rows().view(np.int64).copy()).cuda()
        return t
    A, B, C0 = dev(hA, wa), dev(hB, wn), dev(hC, wn)
    want_leaves = _scheme_leaves(levels)   # by the switch of this process
    out = {}
    old = m4ri_amd.set_max_fuse(0)
    try:
        for fuse in (4, 1):
            m4ri_amd.set_max_fuse(fuse)
            C = C0.clone()
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wa, B.data_ptr(), wn, m, l, n, add=add, cutoff=256)
            st = m4ri_amd.get_stats()
            assert st.levels == levels and (st.leaf_m, st.leaf_l, st.leaf_n) == (m >> levels, l >> levels, n >> levels)
            assert st.leaf_products == (want_leaves if fuse == 4 else 7 ** levels), (fuse, st.leaf_products)
            out[fuse] = C
            assert bool((C[:, hC.rowstride:] == -1).all()), f"max_fuse={fuse}: words of the parent outside C were written"
    finally:
        m4ri_amd.set_max_fuse(old)
    assert torch.equal(out[4], out[1])
    got = Mzd(m, n)
    got.rows()[:, :] = out[4][:, :hC.rowstride].cpu().numpy().view(np.uint64)
    if add:   # C0 ^ A*B: check the product part
        got.rows()[:, :] ^= hC.rows()
    assert freivalds(oracle, hA, hB, got, m, l, n, 79)


@pytest.mark.parametrize("m,l,n,add", [
    (33000, 33000, 33000, False),              # 32768 rows at three levels + 232 rows unsplit; 33000 = 64 * 512 + 232: strips on the inner dimension and the columns
    (20480, 20480, 20480, True),               # 16384 rows at two levels + 4096 rows unsplit, accumulating onto C
])
def test_rows_in_blocks_match_one_product(oracle, m, l, n, add):
    """The engine's own plan cuts rows that do not tile into blocks, each a product of its own at its own depth (engine.hip
    plan_row_blocks); a caller's cutoff means ONE product by the reference's rule.  Same bits, and Freivalds' identity against the
    oracle's thin products."""
    plan = m4ri_amd.plan_row_blocks(m, l, n)
    assert len(plan) == 2 and plan[0][0] in (16384, 32768) and plan[0][1] >= 2 and sum(r for r, _ in plan) == m
    A, B, C0 = dev_random(m, l, 131), dev_random(l, n, 132), dev_random(m, n, 133)
    wl, w = (l + 63) // 64, (n + 63) // 64
    C = C0.clone()
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add)
    assert m4ri_amd.get_stats().levels == plan[0][1]
    D = C0.clone()
    m4ri_amd.mul_dev(D.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add, cutoff=8192)
    torch.cuda.synchronize()
    assert torch.equal(C, D)
    hC = to_host(C, m, n)
    if add:   # C0 + A*B: fold C0 back out before the identity
        hC.rows()[:, :] ^= to_host(C0, m, n).rows()
    assert freivalds(oracle, to_host(A, m, l), to_host(B, l, n), hC, m, l, n, 79)


def test_randomized_shapes_windows_cutoffs(oracle):
    """Seeded fuzz through the C ABI: random shapes, cutoffs, mul/addmul, operands and results that are
    windows of larger parents (column offsets on word boundaries, mzd.c:161), checked word for word
    against the oracle INCLUDING the parents' bits outside the result window."""
    rng = np.random.default_rng(int(os.environ.get("M4RI_AMD_FUZZ_SEED", "20260928")))
    hi = int(os.environ.get("M4RI_AMD_FUZZ_MAXDIM", "1400"))
    for case in range(int(os.environ.get("M4RI_AMD_FUZZ_CASES", "80"))):  # soak runs: raise via the environment
        m, l, n = (int(x) for x in rng.integers(1, hi, 3))
        if case % 9 == 0:
            m, l, n = (int(x) for x in rng.integers(1, 90, 3))          # tiny: the reference's naive fallbacks
        if case % 13 == 0:
            l = int(rng.integers(1, 5)) * 64 * 8                         # multiples of 64*2^L: no strips
        cutoff = int(rng.choice([0, 64, 128, 256, 512, 1024]))
        add = bool(rng.integers(0, 2))

        def operand(rows, cols, seed):
            if rng.integers(0, 2):
                return Mzd.random(rows, cols, seed), None
            pr, pc = rows + int(rng.integers(0, 70)), cols + int(rng.integers(0, 200))
            parent = Mzd.random(pr, pc, seed)
            lowr, lowc = int(rng.integers(0, pr - rows + 1)), int(rng.integers(0, (pc - cols) // 64 + 1)) * 64
            return parent.window(lowr, lowc, lowr + rows, lowc + cols), parent

        A, _ = operand(m, l, 1000 + case)
        B, _ = operand(l, n, 2000 + case)
        C, Cp = operand(m, n, 3000 + case)
        Ch = (Cp.copy() if Cp is not None else C.copy())
        Cw = Ch if Cp is None else Ch.window((C.offset - Cp.offset) // Cp.rowstride, ((C.offset - Cp.offset) % Cp.rowstride) * 64,
                                             (C.offset - Cp.offset) // Cp.rowstride + m, ((C.offset - Cp.offset) % Cp.rowstride) * 64 + n)
        if add:
            m4ri_amd.mzd_addmul(C, A, B, cutoff)
            oracle.addmul(Cw, A, B, cutoff)
        else:
            m4ri_amd.mzd_mul(C, A, B, cutoff)
            oracle.mul(Cw, A, B, cutoff)
        got, want = (Cp if Cp is not None else C), Ch
        assert np.array_equal(got.buf, want.buf), (case, m, l, n, cutoff, add, Cp is not None)


@pytest.mark.parametrize("m,l,n,gen", [
    (464, 16384, 16384 + 37, 1),      # a few rows against a large B: generation 1 by shape (engine.hip pick_leaf)
    (1000, 32768, 32768 + 64, 1),     # up to 1024 rows once l * n >= 2^30
    (1000, 16384, 16384, 4),          # ... below that generation 4 keeps them
    (464, 65536, 4096, 4),            # and with fewer than 16384 columns
])
def test_thin_products_pick_their_leaf_by_shape(oracle, m, l, n, gen):
    hA, hB, hC = Mzd.random(m, l, 141), Mzd.random(l, n, 142), Mzd.random(m, n, 143)
    wl, w = hA.rowstride, hB.rowstride
    A = torch.from_numpy(hA.rows().view(np.int64).copy()).cuda()
    B = torch.from_numpy(hB.rows().view(np.int64).copy()).cuda()
    for add in (False, True):
        C = torch.from_numpy(hC.rows().view(np.int64).copy()).cuda()
        m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add)
        st = m4ri_amd.get_stats()
        assert st.levels == 0 and st.leaf_gen == gen
        want = oracle.addmul(hC.copy(), hA, hB, 0) if add else oracle.mul(None, hA, hB, 0)
        assert to_host(C, m, n).equal(want), f"add={add}"


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("add", [False, True])
def test_leaf_full_rounds_plus_split_tail_matches_plain_launch(add, ragged):
    """300 tiles on 256 CUs: the automatic plan runs one full round unsplit and the 44 remaining tiles
    with their inner dimension split (two launches, atomics only in the tail; engine.hip launch_leaf).
    Same bits as the plain single launch (ksplit = 1)."""
    m, l, n = 3 * 4096, 4096, 100 * 512
    if ragged:  # partial last row tile and last column tile, inner dimension off the word grid
        m, l, n = m - 100, l - 37, n - 77
    A, B = dev_random(m, l, 81), dev_random(l, n, 82)
    C0 = dev_random(m, n, 83)
    w, wl = (n + 63) // 64, (l + 63) // 64
    out = []
    for ksplit in (0, 1):
        C = C0.clone()
        m4ri_amd.m4rm_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add, ksplit=ksplit)
        torch.cuda.synchronize()
        out.append(C)
    assert torch.equal(out[0], out[1])
    assert not torch.equal(out[0], C0)


@pytest.mark.parametrize("add", [False, True])
def test_inner_dimension_splits_fold_to_the_same_bits(add):
    """Explicit inner-dimension splits of a generation-4 launch (slabs + reduce pass; the kernel rounds
    the split count to whole stage pairs, e.g. 32 requested -> 29 used at l = 16453) against ksplit = 1."""
    m, l, n = 2 * 4096 - 11, 16453, 139
    A, B = dev_random(m, l, 91), dev_random(l, n, 92)
    C0 = dev_random(m, n, 93)
    w, wl = (n + 63) // 64, (l + 63) // 64
    ref = None
    for ksplit in (1, 2, 3, 5, 7, 16, 32, 64):
        C = C0.clone()
        m4ri_amd.m4rm_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add, ksplit=ksplit)
        torch.cuda.synchronize()
        if ref is None:
            ref = C
            assert m4ri_amd.get_stats().leaf_gen == 4
        else:
            assert torch.equal(ref, C), f"ksplit={ksplit}"


def test_operands_beyond_4_gib_are_chunked():
    """One operand of a direct (no Strassen level) product larger than the 4 GiB a raw buffer descriptor
    can address: the engine cuts rows of A/C, or the inner dimension, into chunks (engine.hip launch_leaf).
    Checked against the same product assembled from hand-made pieces that are each below the limit."""
    # (a) A = 70000 x 524288 bits = 4.3 GiB, thin B
    m, l, n = 70000, 524288, 512
    A, B = dev_random(m, l, 101), dev_random(l, n, 102)
    wl, w = l // 64, n // 64
    C = torch.empty((m, w), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n)
    assert m4ri_amd.get_stats().levels == 0
    D = torch.empty_like(C)
    h = 32768
    m4ri_amd.mul_dev(D.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, h, l, n)
    m4ri_amd.mul_dev(D.data_ptr() + 8 * h * w, w, A.data_ptr() + 8 * h * wl, wl, B.data_ptr(), w, m - h, l, n)
    torch.cuda.synchronize()
    assert torch.equal(C, D)
    del A, B, C, D
    # (b) B = 524288 x 65536 bits = 4 GiB: inner-dimension chunks, the later ones accumulate
    m, l, n = 4096, 524288, 65536
    A, B = dev_random(m, l, 103), dev_random(l, n, 104)
    wl, w = l // 64, n // 64
    C = torch.empty((m, w), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n)
    assert m4ri_amd.get_stats().levels == 0
    D = torch.empty_like(C)
    h = 262144
    m4ri_amd.mul_dev(D.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, h, n)
    m4ri_amd.mul_dev(D.data_ptr(), w, A.data_ptr() + 8 * (h // 64), wl, B.data_ptr() + 8 * h * w, w, m, l - h, n, add=True)
    torch.cuda.synchronize()
    assert torch.equal(C, D)


def test_schedule_adapts_to_the_memory_that_is_left():
    """The breadth-first schedule of 32768^3 wants ~2.2 GiB of workspace; with all but 0.9 GiB of the HBM
    taken the automatic budget (hipMemGetInfo) sends the top level depth-first (engine.hip product):
    same depth, same bits, no allocation failure."""
    n = 32768
    w = n // 64
    A, B = dev_random(n, n, 111), dev_random(n, n, 112)
    C = torch.empty((n, w), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
    torch.cuda.synchronize()
    assert m4ri_amd.get_stats().levels == 3
    m4ri_amd.lib().m4ri_amd_release_workspace()
    free, _total = torch.cuda.mem_get_info()
    hog = torch.empty(max(0, free - (900 << 20)), dtype=torch.uint8, device="cuda")  # leave ~0.9 GiB
    try:
        D = torch.empty((n, w), dtype=torch.int64, device="cuda")
        m4ri_amd.mul_dev(D.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
        torch.cuda.synchronize()
        assert m4ri_amd.get_stats().levels == 3 and m4ri_amd.get_stats().leaf_launches >= 7
        assert torch.equal(C, D)
    finally:
        del hog
        torch.cuda.empty_cache()


@pytest.mark.parametrize("m,l,n,cutoff,add,budget_mib", [
    (16384, 16384, 16384, 0, False, 64),      # 1 level wanted, its 168 MiB do not fit: depth-first top level
    (8192 + 37, 8192 + 64 + 5, 8192 + 128 + 11, 1024, True, 48),  # 3 levels, ragged, accumulate: depth-first twice
    (32768, 16384, 16384, 0, False, 700),     # rectangular; the sub-products fit breadth-first
])
def test_depth_first_top_levels_when_the_workspace_does_not_fit(m, l, n, cutoff, add, budget_mib):
    """engine.hip product(): levels whose breadth-first workspace exceeds the budget run depth-first (seven
    sub-products one after the other, like strassen.c:111-150); same bits as the all-breadth-first run."""
    wl, w = (l + 63) // 64, (n + 63) // 64
    A, B, C0 = dev_random(m, l, 121), dev_random(l, n, 122), dev_random(m, n, 123)
    ref = C0.clone()
    m4ri_amd.mul_dev(ref.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add, cutoff=cutoff)
    torch.cuda.synchronize()
    levels, launches = m4ri_amd.get_stats().levels, m4ri_amd.get_stats().leaf_launches
    old = m4ri_amd.set_workspace_budget(budget_mib << 20)
    try:
        C = C0.clone()
        m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add, cutoff=cutoff)
        torch.cuda.synchronize()
        st = m4ri_amd.get_stats()
        assert st.levels == levels and st.leaf_launches >= launches + 6  # seven sub-products instead of one batch
        assert torch.equal(C, ref)
    finally:
        m4ri_amd.set_workspace_budget(old)


def test_131072_cubed_vs_reference_fingerprint(oracle):
    """131072^3 (four Strassen levels, 2401 leaves, ~70 GiB of workspace; seeds 7, 8) against the real
    reference's fingerprint (tests/golden/fingerprints_xxl.npz, half an hour of reference CPU time) and
    Freivalds' identity; the depth-first top level gives the same bits."""
    xxl = golden_file("fingerprints_xxl.npz")
    z = np.load(xxl)
    m, l, n = (int(x) for x in z["meta"][0][:3])
    sa, sb = int(z["seeds"][0][0]), int(z["seeds"][0][1])
    A, B = dev_random(m, l, sa), dev_random(l, n, sb)
    C = torch.empty((m, n // 64), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), n // 64, A.data_ptr(), l // 64, B.data_ptr(), n // 64, m, l, n)
    assert m4ri_amd.get_stats().levels == 5
    hC = to_host(C, m, n)
    assert oracle.fingerprint(hC) == int(z["fp"][0])
    assert freivalds(oracle, to_host(A, m, l), to_host(B, l, n), hC, m, l, n, 79)
    old = m4ri_amd.set_workspace_budget(20 << 30)  # 5 levels want ~128 GiB: the top level goes depth-first
    try:
        D = torch.empty_like(C)
        m4ri_amd.mul_dev(D.data_ptr(), n // 64, A.data_ptr(), l // 64, B.data_ptr(), n // 64, m, l, n)
        assert torch.equal(C, D)
    finally:
        m4ri_amd.set_workspace_budget(old)


def test_products_on_different_streams_share_the_workspace_safely():
    """Two streams, products issued alternately without any host synchronisation: the engine orders them on
    the device (one workspace per device), so every result equals the single-stream one."""
    n = 16384
    w = n // 64
    A, B = dev_random(n, n, 131), dev_random(n, n, 132)
    A2, B2 = dev_random(n, n, 133), dev_random(n, n, 134)
    ref1, ref2 = torch.empty((n, w), dtype=torch.int64, device="cuda"), torch.empty((n, w), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(ref1.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
    m4ri_amd.mul_dev(ref2.data_ptr(), w, A2.data_ptr(), w, B2.data_ptr(), w, n, n, n)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for k in range(6):
        C = torch.empty((n, w), dtype=torch.int64, device="cuda")
        st, (a, b) = ((s1, (A, B)) if k % 2 == 0 else (s2, (A2, B2)))
        m4ri_amd.mul_dev(C.data_ptr(), w, a.data_ptr(), w, b.data_ptr(), w, n, n, n, stream=st.cuda_stream)
        outs.append(C)
    torch.cuda.synchronize()
    for k, C in enumerate(outs):
        assert torch.equal(C, ref1 if k % 2 == 0 else ref2), k


@pytest.mark.skipif(os.environ.get("M4RI_AMD_HUGE", "") == "", reason="8 GiB per matrix, minutes of host time: set M4RI_AMD_HUGE=1")
def test_262144_cubed_vs_reference_fingerprint(oracle):
    """262144^3 (8 GiB per matrix; five Strassen levels, the top two depth-first; seeds 9, 10) against the
    real reference's fingerprint (tests/golden/fingerprints_huge.npz, make_golden.py --huge)."""
    huge = golden_file("fingerprints_huge.npz")
    z = np.load(huge)
    m, l, n = (int(x) for x in z["meta"][0][:3])
    A, B = dev_random(m, l, int(z["seeds"][0][0])), dev_random(l, n, int(z["seeds"][0][1]))
    C = torch.empty((m, n // 64), dtype=torch.int64, device="cuda")
    m4ri_amd.mul_dev(C.data_ptr(), n // 64, A.data_ptr(), l // 64, B.data_ptr(), n // 64, m, l, n)
    assert m4ri_amd.get_stats().levels == 6
    assert oracle.fingerprint(to_host(C, m, n)) == int(z["fp"][0])


def test_large_ragged_shapes_vs_reference_fingerprints(oracle):
    """Large products off every grid (all three remainder strips, partly filled tiles, an inner dimension
    that is not a multiple of 64, one accumulate) against the real reference's fingerprints
    (tests/golden/fingerprints_ragged_xl.npz, make_golden.py --ragged-xl)."""
    path = golden_file("fingerprints_ragged_xl.npz")
    z = np.load(path)
    for op, (m, l, n, par), (sa, sb, sc), fp in zip(z["ops"], z["meta"], z["seeds"], z["fp"]):
        m, l, n, par = int(m), int(l), int(n), int(par)
        wl, w = (l + 63) // 64, (n + 63) // 64
        A, B = dev_random(m, l, int(sa)), dev_random(l, n, int(sb))
        add = str(op) == "addmul"
        C = dev_random(m, n, int(sc)) if add else torch.empty((m, w), dtype=torch.int64, device="cuda")
        if str(op) == "m4rm":
            m4ri_amd.m4rm_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n)
        else:
            m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, m, l, n, add=add, cutoff=par)
        hC = to_host(C, m, n)
        assert oracle.fingerprint(hC) == int(fp), (str(op), m, l, n)
        assert sha_matches(hC, str(op), m, l, n, int(sa), par), (str(op), m, l, n)
        del A, B, C, hC


@pytest.mark.parametrize("pinned", [False, True])
def test_large_windowed_addmul_vs_reference_parent_fingerprint(oracle, pinned):
    """C_window += A_window * B_window on windows of three 30000 x 30000 parents (tests/golden/make_golden.py
    WINDOW_XL): the fingerprint of the WHOLE parent of C must be that of the reference's product of clean
    copies of the windows, merged under the column mask -- through the host entry point from host memory,
    and with the three parents pinned (windows used in place on the device).  (The reference applied to the
    windows themselves differs in the last word column: its < 54-column fallback mishandles a windowed B
    with non-zero excess; that fingerprint is stored too and must NOT be what we produce.)"""
    path = golden_file("fingerprint_window_xl.npz")
    W = dict(pa=(30000, 30000, 41), pb=(30000, 30000, 42), pc=(30000, 30000, 43),
             a=(100, 64, 20100, 16448 + 37), b=(7, 128, 7 + 16384 + 37, 128 + 21000 + 5), c=(9000, 6400, 29000, 6400 + 21000 + 5))
    Pa, Pb, Pc = (Mzd.random(*W[k]) for k in ("pa", "pb", "pc"))
    A, B, C = Pa.window(*W["a"]), Pb.window(*W["b"]), Pc.window(*W["c"])
    if pinned:
        for P in (Pa, Pb, Pc):
            m4ri_amd.pin(P)
    m4ri_amd.mzd_addmul(C, A, B, 0)
    if pinned:
        for P in (Pa, Pb, Pc):
            m4ri_amd.unpin(P)
    z = np.load(path)
    got = oracle.fingerprint(Pc)
    assert got == int(z["fp"][0])
    assert got != int(z["fp_reference_on_windows"][0])
    assert sha_matches(Pc, "window_addmul_parent", A.nrows, A.ncols, B.ncols, 41)
