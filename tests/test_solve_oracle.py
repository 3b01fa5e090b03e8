"""The oracle's restatements of the drivers over PLUQ pinned against the reference: mzd_apply_p_left{,_trans}
(m4ri/mzp.c:65-81), mzd_solve_left / mzd_pluq_solve_left (m4ri/solve.c:30-152; A is left holding its PLUQ, B the solution
with the undefined rows zero, -1 = inconsistent), mzd_kernel_left_pluq (solve.c:154-191) and mzd_inv_m4ri
(m4ri/brilliantrussian.c:971-997)."""
import numpy as np
import pytest

from m4ri_amd.mzd import Mzd
from test_ple_oracle import _defects, _make

SYSTEMS = [(1, 1, 1), (5, 5, 3), (64, 64, 64), (65, 63, 10), (63, 65, 70), (100, 100, 130), (200, 70, 33), (70, 200, 129), (300, 300, 64), (513, 511, 200),
           (257, 600, 65)]


@pytest.mark.parametrize("m,n", [(1, 1), (3, 64), (10, 65), (70, 130), (200, 333), (1000, 64)])
def test_apply_p_left_matches_reference(oracle, reference, m, n):
    rng = np.random.default_rng(m * 7 + n)
    P = np.array([rng.integers(i, m) for i in range(m)], dtype=np.int32)
    A = Mzd.random(m, n, 5)
    for trans, which in ((False, "mzd_apply_p_left"), (True, "mzd_apply_p_left_trans")):
        Ao, Ar = A.copy(), A.copy()
        oracle.apply_p_left(Ao, P, trans)
        reference.apply_p(Ar, P, which)
        assert np.array_equal(Ar.valid_words(), Ao.valid_words()), which


@pytest.mark.parametrize("m,n,k", SYSTEMS)
@pytest.mark.parametrize("kind", ["random", "lowrank", "zerocols"])
@pytest.mark.parametrize("check", [False, True])
def test_solve_left_matches_reference(oracle, reference, m, n, k, kind, check):
    """Consistent systems (B = A X) and, with the check, inconsistent ones (random B)."""
    A = _make(kind, m, n, 6000 + 7 * m + n)
    rows = max(m, n)
    for consistent in (True, False):
        B = Mzd(rows, k)
        if consistent:
            X = Mzd.random(n, k, 77)
            B.valid_words()[:m] = oracle.mul(None, A, X, 0).valid_words()
        else:
            B.valid_words()[:m] = Mzd.random(m, k, 78).valid_words()
        Ao, Ar, Bo, Br = A.copy(), A.copy(), B.copy(), B.copy()
        ro, rr = oracle.solve_left(Ao, Bo, check), reference.solve_left(Ar, Br, check)
        assert ro == rr, (consistent, ro, rr)
        assert np.array_equal(Ar.valid_words(), Ao.valid_words()), "A (its PLUQ) differs"
        assert np.array_equal(Br.valid_words(), Bo.valid_words()), "B differs"
        if consistent:
            assert ro == 0
            assert np.array_equal(oracle.mul(None, A, _top(Bo, n, k), 0).valid_words(), B.valid_words()[:m])


def _top(B, n, k):
    X = Mzd(n, k)
    X.valid_words()[:] = B.valid_words()[:n]
    return X


@pytest.mark.parametrize("m,n", [(5, 5), (64, 64), (65, 63), (63, 65), (100, 300), (300, 100), (513, 511), (200, 1000)])
@pytest.mark.parametrize("kind", ["random", "lowrank", "zerocols"])
def test_kernel_left_pluq_matches_reference(oracle, reference, m, n, kind):
    A = _make(kind, m, n, 7000 + 7 * m + n)
    Ao, Ar = A.copy(), A.copy()
    r, Ro = oracle.kernel_left_pluq(Ao)
    Rr = reference.kernel_left_pluq(Ar)
    assert (Ro is None) == (Rr is None)
    assert np.array_equal(Ar.valid_words(), Ao.valid_words())
    if Ro is not None:
        assert (Rr.nrows, Rr.ncols) == (n, n - r) and np.array_equal(Rr.valid_words(), Ro.valid_words())
        assert not oracle.mul(None, A, Ro, 0).valid_words().any(), "A * R != 0"


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 100, 128, 200, 513])
def test_inverse_matches_reference(oracle, reference, n):
    for seed in (1, 2):
        A = Mzd.random(n, n, 8000 + n + seed)
        if seed == 2 and n > 2:
            A.valid_words()[n // 2] = A.valid_words()[0]  # singular: the reference returns what the elimination leaves
        assert np.array_equal(reference.inv(A).valid_words(), oracle.inv(A).valid_words()), (n, seed)


def test_larger_system_matches_reference(oracle, reference):
    """Above the PLE recursion cutoff, rank deficient: the PLUQ left in A carries the reference's leftovers in Q."""
    m, n, k = 4200, 8256, 100
    A = _defects(m, n, 99)
    X = Mzd.random(n, k, 5)
    B = Mzd(n, k)
    B.valid_words()[:m] = oracle.mul(None, A, X, 0).valid_words()
    Ao, Ar, Bo, Br = A.copy(), A.copy(), B.copy(), B.copy()
    assert oracle.solve_left(Ao, Bo, True) == reference.solve_left(Ar, Br, True) == 0
    assert np.array_equal(Ar.valid_words(), Ao.valid_words()) and np.array_equal(Br.valid_words(), Bo.valid_words())


def test_padding_rows_of_b_match_reference(oracle, reference):
    """The pre-check of _mzd_solve_left looks at B from row m + 1 on (solve.c:125), not from row m."""
    m, n, k = 100, 160, 70
    A = Mzd.random(m, n, 1)
    for bad_row in (m, m + 1, n - 1):
        B = Mzd(n, k)
        B.valid_words()[:m] = oracle.mul(None, A, Mzd.random(n, k, 77), 0).valid_words()
        B.valid_words()[bad_row, 0] = np.uint64(5)
        Ao, Bo, Ar, Br = A.copy(), B.copy(), A.copy(), B.copy()
        assert oracle.solve_left(Ao, Bo, True) == reference.solve_left(Ar, Br, True)
        assert np.array_equal(Ar.valid_words(), Ao.valid_words()) and np.array_equal(Br.valid_words(), Bo.valid_words())
