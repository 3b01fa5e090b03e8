"""The oracle's echelon forms (plain Gauss / Gauss-Jordan, first-row pivots) pinned against the reference's three
drivers -- mzd_echelonize_m4ri for several table widths k (m4ri/brilliantrussian.c:603-841: strips of 6k columns,
Gray-code tables), mzd_echelonize_pluq (m4ri/echelonform.c:37-139: PLE / PLUQ, triangular solve, column permutation) and
mzd_echelonize (the density heuristic that switches between them): same rank, same matrix, reduced (full = 1) and not
(full = 0), on full-rank, rank-deficient, tall, wide, sparse and structured inputs."""
import numpy as np
import pytest

from m4ri_amd.mzd import Mzd
from test_ple_oracle import SHAPES, _defects, _make


@pytest.mark.parametrize("m,n", SHAPES)
@pytest.mark.parametrize("kind", ["random", "lowrank", "sparse", "zerocols"])
@pytest.mark.parametrize("full", [0, 1])
def test_echelon_forms_match_reference(oracle, reference, m, n, kind, full):
    A = _make(kind, m, n, 4000 + 7 * m + n)
    Ao = A.copy()
    want = oracle.echelonize(Ao, full)
    for which, k in (("mzd_echelonize_m4ri", 0), ("mzd_echelonize_m4ri", 1), ("mzd_echelonize_m4ri", 3), ("mzd_echelonize_m4ri", 8),
                     ("mzd_echelonize_pluq", 0), ("mzd_echelonize", 0), ("mzd_echelonize_naive", 0)):
        Ar = A.copy()
        assert reference.echelonize(Ar, full, which, k) == want, (which, k)
        assert np.array_equal(Ar.valid_words(), Ao.valid_words()), (which, k)


@pytest.mark.parametrize("m,n,dup,zero", [(4200, 8256, 0, 0), (9000, 4200, 300, 100), (2500, 2500, 0, 0)])
@pytest.mark.parametrize("full", [0, 1])
def test_larger_echelon_forms_match_reference(oracle, reference, m, n, dup, zero, full):
    """Sizes where mzd_echelonize's heuristic and the recursive PLE come into play, with scattered pivot-free columns."""
    A = _defects(m, n, 5000 + m + n, dup, zero)
    Ao = A.copy()
    want = oracle.echelonize(Ao, full)
    for which in ("mzd_echelonize_m4ri", "mzd_echelonize_pluq", "mzd_echelonize"):
        Ar = A.copy()
        assert reference.echelonize(Ar, full, which) == want, which
        assert np.array_equal(Ar.valid_words(), Ao.valid_words()), which


@pytest.mark.parametrize("m,n", [(1, 1), (3, 64), (10, 65), (70, 130), (200, 333), (64, 1000)])
def test_apply_p_right_matches_reference(oracle, reference, m, n):
    rng = np.random.default_rng(m * 7 + n)
    Q = np.array([rng.integers(i, n) for i in range(n)], dtype=np.int32)
    A = Mzd.random(m, n, 5)
    for trans, which in ((False, "mzd_apply_p_right"), (True, "mzd_apply_p_right_trans")):
        Ao, Ar = A.copy(), A.copy()
        oracle.apply_p_right(Ao, Q, trans)
        reference.apply_p(Ar, Q, which)
        assert np.array_equal(Ar.valid_words(), Ao.valid_words()), which
