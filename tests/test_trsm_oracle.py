"""The oracle's triangular solves (plain substitution) pinned against the real reference's recursive and
Four-Russians schedules (m4ri/triangular.c:396-514, m4ri/triangular_russian.c:50-330): same bits, for
triangles that carry garbage in the diagonal and the other triangle (never read), right-hand sides that
are windows with non-zero excess, and sizes on both sides of the reference's 64 / 2048-row switch points
(shapes of the reference's tests/test_trsm.c: m, n up to 2 * MAXSIZE around multiples of 64)."""
import numpy as np
import pytest

from m4ri_amd.mzd import Mzd

SHAPES = [(1, 1), (2, 65), (57, 10), (64, 64), (65, 1), (100, 300), (128, 64), (200, 513), (511, 129), (1000, 70), (2049, 200), (2500, 131)]


@pytest.mark.parametrize("mb,nb", SHAPES)
@pytest.mark.parametrize("upper", [False, True])
def test_trsm_matches_reference(oracle, reference, mb, nb, upper):
    T = Mzd.random(mb, mb, 100 + mb)           # full random matrix: diagonal and the other triangle are junk
    B = Mzd.random(mb, nb, 200 + nb)
    Bo, Br, Br2 = B.copy(), B.copy(), B.copy()
    if upper:
        oracle.trsm_upper_left(T, Bo)
        reference.L.mzd_trsm_upper_left(T.ptr, Br.ptr, 0)
        reference.L._mzd_trsm_upper_left_russian(T.ptr, Br2.ptr, 0)
    else:
        oracle.trsm_lower_left(T, Bo)
        reference.L.mzd_trsm_lower_left(T.ptr, Br.ptr, 0)
        reference.L._mzd_trsm_lower_left_russian(T.ptr, Br2.ptr, 0)
    assert Bo.equal(Br) and Bo.equal(Br2)
    assert np.array_equal(Bo.valid_words(), Br.valid_words())


@pytest.mark.parametrize("upper", [False, True])
def test_trsm_on_windows(oracle, reference, upper):
    """T a window into a larger matrix (as PLE hands A00 to TRSM, ple.c:123-125), B a window with non-zero
    excess inside a pattern-filled parent: the parent outside the window is untouched."""
    P, Q = Mzd.random(700, 900, 7), Mzd.random(600, 1100, 8)
    for (mb, nb, c0) in [(300, 333, 64), (130, 65, 128), (513, 700, 0)]:
        T = P.window(10, 64, 10 + mb, 64 + mb)
        Qo, Qr = Mzd(600, 1100, buf=Q.buf.copy()), Mzd(600, 1100, buf=Q.buf.copy())
        bo, br = Qo.window(5, c0, 5 + mb, c0 + nb), Qr.window(5, c0, 5 + mb, c0 + nb)
        if upper:
            oracle.trsm_upper_left(T, bo)
            reference.L.mzd_trsm_upper_left(T.ptr, br.ptr, 0)
        else:
            oracle.trsm_lower_left(T, bo)
            reference.L.mzd_trsm_lower_left(T.ptr, br.ptr, 0)
        assert np.array_equal(Qo.buf, Qr.buf)
