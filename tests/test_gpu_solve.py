"""The drivers over PLUQ on the GPU (include/m4ri_amd.h: mzd_solve_left, _mzd_solve_left, mzd_pluq_solve_left,
_mzd_pluq_solve_left, mzd_kernel_left_pluq, mzd_inv_m4ri, mzd_apply_p_left{,_trans}; reference m4ri/solve.c:30-191,
m4ri/brilliantrussian.c:971-997, m4ri/mzp.c:65-81) against the oracle's restatements, which tests/test_solve_oracle.py pins
to the reference: every output matrix bit for bit, return values included."""
import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd
from test_ple_oracle import _defects, _make
from test_solve_oracle import SYSTEMS, _top

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)


@pytest.mark.parametrize("m,n", [(1, 1), (3, 64), (10, 65), (70, 130), (200, 333), (1000, 64), (5000, 700)])
def test_apply_p_left(oracle, m, n):
    rng = np.random.default_rng(m * 7 + n)
    P = np.array([rng.integers(i, m) for i in range(m)], dtype=np.int32)
    A = Mzd.random(m, n, 5)
    for trans in (False, True):
        Ao, Ag = A.copy(), A.copy()
        oracle.apply_p_left(Ao, P, trans)
        m4ri_amd.mzd_apply_p_left(Ag, P, trans)
        assert np.array_equal(Ag.valid_words(), Ao.valid_words()), trans


def _system(oracle, A, rows, k, consistent):
    B = Mzd(rows, k)
    if consistent:
        B.valid_words()[:A.nrows] = oracle.mul(None, A, Mzd.random(A.ncols, k, 77), 0).valid_words()
    else:
        B.valid_words()[:A.nrows] = Mzd.random(A.nrows, k, 78).valid_words()
    return B


@pytest.mark.parametrize("m,n,k", SYSTEMS)
@pytest.mark.parametrize("kind", ["random", "lowrank", "zerocols"])
@pytest.mark.parametrize("check", [False, True])
def test_solve_left_matches_oracle(oracle, m, n, k, kind, check):
    A = _make(kind, m, n, 6000 + 7 * m + n)
    for consistent in (True, False):
        B = _system(oracle, A, max(m, n), k, consistent)
        Ao, Bo = A.copy(), B.copy()
        want = oracle.solve_left(Ao, Bo, check)
        for which in ("mzd_solve_left", "_mzd_solve_left"):
            Ag, Bg = A.copy(), B.copy()
            assert m4ri_amd.mzd_solve_left(Ag, Bg, 0, check, which) == want, which
            assert np.array_equal(Ag.valid_words(), Ao.valid_words()), which + ": A"
            assert np.array_equal(Bg.valid_words(), Bo.valid_words()), which + ": B"
        # the same from a decomposition made beforehand
        Ad = A.copy()
        r, P, Q = m4ri_amd.mzd_ple(Ad, 0, "_mzd_pluq")
        for which in ("mzd_pluq_solve_left", "_mzd_pluq_solve_left"):
            Bg = B.copy()
            assert m4ri_amd.mzd_pluq_solve_left(Ad, r, P, Q, Bg, 0, check, which) == want, which
            assert np.array_equal(Bg.valid_words(), Bo.valid_words()), which


@pytest.mark.parametrize("m,n,k,dup,zero", [(4200, 8256, 100, 0, 0), (9000, 4200, 300, 300, 100), (3000, 3000, 3000, 0, 0), (6000, 2000, 64, 0, 0)])
def test_larger_systems_match_oracle(oracle, m, n, k, dup, zero):
    A = _defects(m, n, 99 + m, dup, zero)
    B = _system(oracle, A, max(m, n), k, True)
    Ao, Bo, Ag, Bg = A.copy(), B.copy(), A.copy(), B.copy()
    assert oracle.solve_left(Ao, Bo, True) == 0 and m4ri_amd.mzd_solve_left(Ag, Bg, 0, True) == 0
    assert np.array_equal(Ag.valid_words(), Ao.valid_words()) and np.array_equal(Bg.valid_words(), Bo.valid_words())
    assert np.array_equal(m4ri_amd.mzd_mul(None, A, _top(Bg, n, k), 0).valid_words(), B.valid_words()[:m])


def test_solution_at_scale():
    """n = 20000, rank deficient, where the oracle is too slow: A X == B through the product, undefined rows zero."""
    m, n, k = 20000, 24000, 500
    A = m4ri_amd.mzd_mul(None, Mzd.random(m, 9000, 1), Mzd.random(9000, n, 2), 0)
    B = Mzd(n, k)
    B.valid_words()[:m] = m4ri_amd.mzd_mul(None, A, Mzd.random(n, k, 3), 0).valid_words()
    Ag, Bg = A.copy(), B.copy()
    assert m4ri_amd.mzd_solve_left(Ag, Bg, 0, True) == 0
    assert np.array_equal(m4ri_amd.mzd_mul(None, A, _top(Bg, n, k), 0).valid_words(), B.valid_words()[:m])
    Bbad = B.copy()
    Bbad.valid_words()[:m] ^= Mzd.random(m, k, 4).valid_words()
    assert m4ri_amd.mzd_solve_left(A.copy(), Bbad, 0, True) == -1


@pytest.mark.parametrize("m,n", [(5, 5), (64, 64), (65, 63), (63, 65), (100, 300), (300, 100), (513, 511), (200, 1000), (3000, 5000), (0, 7), (7, 0)])
@pytest.mark.parametrize("kind", ["random", "lowrank", "zerocols"])
def test_kernel_left_pluq_matches_oracle(oracle, m, n, kind):
    A = _make(kind, m, n, 7000 + 7 * m + n) if m and n else Mzd(m, n)
    Ao, Ag = A.copy(), A.copy()
    if m and n:
        r, Ro = oracle.kernel_left_pluq(Ao)
    else:
        Ro = None if n == 0 else Mzd.from_bits(np.eye(n, dtype=np.uint8))
    Rg = m4ri_amd.mzd_kernel_left_pluq(Ag)
    assert (Ro is None) == (Rg is None)
    assert np.array_equal(Ag.valid_words(), Ao.valid_words())
    if Ro is not None:
        assert (Rg.nrows, Rg.ncols) == (Ro.nrows, Ro.ncols) and np.array_equal(Rg.valid_words(), Ro.valid_words())


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 100, 128, 200, 513, 3000])
def test_inverse_matches_oracle(oracle, n):
    for seed in (1, 2):
        A = Mzd.random(n, n, 8000 + n + seed)
        if seed == 2 and n > 2:
            A.valid_words()[n // 2] = A.valid_words()[0]
        want = oracle.inv(A)
        assert np.array_equal(m4ri_amd.mzd_inv_m4ri(A).valid_words(), want.valid_words()), (n, seed)
        Bpre = Mzd.random(n, n, 3)
        m4ri_amd.mzd_inv_m4ri(A, Bpre)
        assert np.array_equal(Bpre.valid_words(), want.valid_words())


def test_inverse_at_scale():
    n = 16384
    for seed in range(1, 40):  # a random matrix over GF(2) is invertible with probability 0.29
        A = Mzd.random(n, n, 12345 + seed)
        if m4ri_amd.mzd_echelonize(A.copy(), 0) == n:
            break
    else:
        pytest.fail("no invertible matrix among 39 seeds")
    P = m4ri_amd.mzd_mul(None, A, m4ri_amd.mzd_inv_m4ri(A), 0)
    w = P.valid_words()
    eye = np.zeros_like(w)
    idx = np.arange(n)
    eye[idx, idx // 64] = np.uint64(1) << (idx % 64).astype(np.uint64)
    assert np.array_equal(w, eye)


def test_padding_rows_of_b(oracle):
    """m < n: B has n rows.  With the check, rows m+1 .. n-1 must be zero on entry (the reference looks from row m+1 on,
    solve.c:125 -- row m itself is not looked at) or the call returns -1 and touches nothing."""
    m, n, k = 100, 160, 70
    A = Mzd.random(m, n, 1)
    for bad_row in (m, m + 1, n - 1):
        B = _system(oracle, A, n, k, True)
        B.valid_words()[bad_row, 0] = np.uint64(5)
        Ao, Bo, Ag, Bg = A.copy(), B.copy(), A.copy(), B.copy()
        want = oracle.solve_left(Ao, Bo, True)
        assert m4ri_amd.mzd_solve_left(Ag, Bg, 0, True) == want == (0 if bad_row == m else -1)
        assert np.array_equal(Ag.valid_words(), Ao.valid_words()) and np.array_equal(Bg.valid_words(), Bo.valid_words())
        if want == -1:
            assert np.array_equal(Ag.valid_words(), A.valid_words()) and np.array_equal(Bg.valid_words(), B.valid_words())
