"""Triangular solves on the GPU (include/m4ri_amd.h: mzd_trsm_{lower,upper}_left, their _mzd_ and _russian
forms, reference m4ri/triangular.c:396-514 and m4ri/triangular_russian.c:50-330) against the oracle's
substitution (pinned to the reference in test_trsm_oracle.py).  The solution is unique: bit-exact."""
import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)


SHAPES = [(1, 1), (2, 65), (57, 10), (64, 64), (65, 1), (100, 300), (128, 64), (200, 513), (511, 129), (1000, 70), (2049, 200),
          (2500, 131), (4096, 4096), (3000, 10000)]


@pytest.mark.parametrize("mb,nb", SHAPES)
@pytest.mark.parametrize("upper", [False, True])
def test_trsm_matches_oracle(oracle, mb, nb, upper):
    T = Mzd.random(mb, mb, 100 + mb)  # diagonal and the other triangle are junk, never read
    B = Mzd.random(mb, nb, 200 + nb)
    want = (oracle.trsm_upper_left if upper else oracle.trsm_lower_left)(T, B.copy())
    L = m4ri_amd.lib()
    names = ("mzd_trsm_upper_left", "_mzd_trsm_upper_left", "_mzd_trsm_upper_left_russian") if upper else \
            ("mzd_trsm_lower_left", "_mzd_trsm_lower_left", "_mzd_trsm_lower_left_russian")
    for name in names:
        got = B.copy()
        getattr(L, name)(T.ptr, got.ptr, 0)
        assert np.array_equal(got.valid_words(), want.valid_words()), (name, mb, nb)


@pytest.mark.parametrize("upper", [False, True])
def test_trsm_on_windows_keeps_the_parent(oracle, upper):
    P, Q = Mzd.random(700, 900, 7), Mzd.random(600, 1100, 8)
    for (mb, nb, c0) in [(300, 333, 64), (130, 65, 128), (513, 700, 0)]:
        T = P.window(10, 64, 10 + mb, 64 + mb)
        Qo, Qg = Mzd(600, 1100, buf=Q.buf.copy()), Mzd(600, 1100, buf=Q.buf.copy())
        bo, bg = Qo.window(5, c0, 5 + mb, c0 + nb), Qg.window(5, c0, 5 + mb, c0 + nb)
        if upper:
            oracle.trsm_upper_left(T, bo)
            m4ri_amd.mzd_trsm_upper_left(T, bg)
        else:
            oracle.trsm_lower_left(T, bo)
            m4ri_amd.mzd_trsm_lower_left(T, bg)
        assert np.array_equal(Qo.buf, Qg.buf)


def test_trsm_on_pinned_matrices(oracle):
    """The PLE pattern (ple.c:123-126): A01 <- A00^-1 A01, then A11 += A10 * A01, all windows of ONE pinned matrix."""
    n, r1, n1 = 1500, 640, 704
    A = Mzd.random(n, n, 51)
    Ah = A.copy()
    m4ri_amd.pin(A)

    def blocks(M):
        return M.window(0, 0, r1, r1), M.window(0, n1, r1, n), M.window(r1, 0, n, r1), M.window(r1, n1, n, n)

    a00, a01, a10, a11 = blocks(A)
    h00, h01, h10, h11 = blocks(Ah)
    m4ri_amd.mzd_trsm_lower_left(a00, a01)
    m4ri_amd.mzd_addmul(a11, a10, a01, 0)
    oracle.trsm_lower_left(h00, h01)
    oracle.addmul(h11, h10, h01, 0)
    m4ri_amd.unpin(A)
    assert np.array_equal(A.rows(), Ah.rows())


def test_large_solve_is_an_inverse(oracle):
    """16384-row solve against a wide right-hand side: L * X == B through the (independently tested) product."""
    mb, nb = 16384, 20000
    rng = Mzd.random(mb, mb, 61)
    bits_lower = np.tril(np.ones((64, 64), dtype=np.uint8))  # build a clean unit lower triangular L word by word
    L = Mzd.init(mb, mb)
    w = rng.valid_words()
    rows = np.arange(mb)
    for j in range(L.width):  # word j of row r keeps bits < r - 64 j (all of it when r >= 64 (j + 1)), plus the diagonal
        keep = np.clip(rows - 64 * j, 0, 64).astype(np.uint64)
        mask = np.where(keep >= 64, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << keep) - np.uint64(1))
        diag = np.where((rows // 64) == j, np.uint64(1) << (rows % 64).astype(np.uint64), np.uint64(0))
        L.valid_words()[:, j] = (w[:, j] & mask) | diag
    B = Mzd.random(mb, nb, 62)
    X = B.copy()
    m4ri_amd.mzd_trsm_lower_left(L, X)
    assert m4ri_amd.mzd_mul(None, L, X, 0).equal(B)
    U = Mzd.random(mb, mb, 63)
    Y = B.copy()
    m4ri_amd.mzd_trsm_upper_left(U, Y)
    # U's junk lower triangle must not matter: solve again with it cleared
    Uc = Mzd.init(mb, mb)
    for j in range(U.width):
        keep = np.clip(rows - 64 * j + 1, 0, 64).astype(np.uint64)   # bits <= r - 64 j are at or below the diagonal
        low = np.where(keep >= 64, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << keep) - np.uint64(1))
        diag = np.where((rows // 64) == j, np.uint64(1) << (rows % 64).astype(np.uint64), np.uint64(0))
        Uc.valid_words()[:, j] = (U.valid_words()[:, j] & ~low) | diag
    assert m4ri_amd.mzd_mul(None, Uc, Y, 0).equal(B)


RIGHT_SHAPES = [(1, 1), (65, 2), (10, 57), (64, 64), (1, 65), (300, 100), (64, 128), (513, 200), (129, 511), (70, 1000), (200, 2049),
                (131, 2500), (400, 4096)]


def _unit_diag(T):
    for i in range(T.nrows):
        T.valid_words()[i, i // 64] |= np.uint64(1) << np.uint64(i % 64)
    return T


@pytest.mark.parametrize("mb,nb", RIGHT_SHAPES)
@pytest.mark.parametrize("upper", [False, True])
def test_right_trsm_matches_oracle(oracle, mb, nb, upper):
    """B <- B T^-1 (mzd_trsm_{upper,lower}_right and their _mzd_ forms; m4ri/triangular.c:41-130, :301-393)."""
    T = _unit_diag(Mzd.random(nb, nb, 300 + nb))
    B = Mzd.random(mb, nb, 400 + mb)
    want = (oracle.trsm_upper_right if upper else oracle.trsm_lower_right)(T, B.copy())
    L = m4ri_amd.lib()
    for name in (("mzd_trsm_upper_right", "_mzd_trsm_upper_right") if upper else ("mzd_trsm_lower_right", "_mzd_trsm_lower_right")):
        got = B.copy()
        getattr(L, name)(T.ptr, got.ptr, 0)
        assert np.array_equal(got.valid_words(), want.valid_words()), (name, mb, nb)


@pytest.mark.parametrize("upper", [False, True])
def test_right_trsm_on_windows_keeps_the_parent(oracle, upper):
    P, Q = _unit_diag(Mzd.random(900, 900, 17)), Mzd.random(600, 1100, 18)
    for (mb, nb, c0) in [(300, 333, 64), (130, 65, 128), (513, 700, 0)]:
        T = P.window(64, 64, 64 + nb, 64 + nb)   # the window's diagonal is the parent's
        Qo, Qg = Mzd(600, 1100, buf=Q.buf.copy()), Mzd(600, 1100, buf=Q.buf.copy())
        bo, bg = Qo.window(5, c0, 5 + mb, c0 + nb), Qg.window(5, c0, 5 + mb, c0 + nb)
        (oracle.trsm_upper_right if upper else oracle.trsm_lower_right)(T, bo)
        getattr(m4ri_amd.lib(), "mzd_trsm_upper_right" if upper else "mzd_trsm_lower_right")(T.ptr, bg.ptr, 0)
        assert np.array_equal(Qo.buf, Qg.buf)


def test_large_right_solve_is_an_inverse():
    """20000 x 8192 right-hand solves: X * T == B through the (independently tested) product, T's junk triangle cleared."""
    mb, nb = 20000, 8192
    B = Mzd.random(mb, nb, 71)
    rows = np.arange(nb)
    R = Mzd.random(nb, nb, 72)
    for upper in (True, False):
        Tc = Mzd.init(nb, nb)
        for j in range(Tc.width):
            keep = np.clip(rows - 64 * j + (1 if upper else 0), 0, 64).astype(np.uint64)   # bits <= r (upper) / < r (lower) of word j
            low = np.where(keep >= 64, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << keep) - np.uint64(1))
            diag = np.where((rows // 64) == j, np.uint64(1) << (rows % 64).astype(np.uint64), np.uint64(0))
            w = R.valid_words()[:, j]
            Tc.valid_words()[:, j] = ((w & ~low) if upper else (w & low)) | diag
        X = B.copy()
        (m4ri_amd.mzd_trsm_upper_right if False else getattr(m4ri_amd.lib(), "mzd_trsm_upper_right" if upper else "mzd_trsm_lower_right"))(Tc.ptr, X.ptr, 0)
        assert m4ri_amd.mzd_mul(None, X, Tc, 0).equal(B), "upper" if upper else "lower"


BIG_SHAPES = [(4097, 200), (5000, 300), (8192, 64), (9000, 100), (12289, 70), (8200, 8300)]


@pytest.mark.parametrize("mb,nb", BIG_SHAPES)
@pytest.mark.parametrize("upper", [False, True])
def test_trsm_big_shapes(oracle, mb, nb, upper):
    """Systems of more than 4096 rows: the solver works with the inverses of 4096-row blocks (trsm.hip: build_big_inverses),
    the last block ragged (test_trsm_big_blocks_off repeats the file with the 512-row blocks only)."""
    T = Mzd.random(mb, mb, 300 + mb)
    B = Mzd.random(mb, nb, 400 + nb)
    want = (oracle.trsm_upper_left if upper else oracle.trsm_lower_left)(T, B.copy())
    got = B.copy()
    (m4ri_amd.mzd_trsm_upper_left if upper else m4ri_amd.mzd_trsm_lower_left)(T, got)
    assert np.array_equal(got.valid_words(), want.valid_words())


@pytest.mark.parametrize("mb,nb", [(16, 4097), (20, 5000), (8, 8192), (12, 9000)])  # few rows: the oracle substitutes column by column
@pytest.mark.parametrize("upper", [False, True])
def test_right_trsm_big_shapes(oracle, mb, nb, upper):
    """Right-hand solves against triangles of more than 4096 columns: the same 4096-row block inverses, used from the right."""
    T = Mzd.random(nb, nb, 500 + nb)
    idx = np.arange(nb)
    T.valid_words()[idx, idx // 64] |= np.uint64(1) << (idx % 64).astype(np.uint64)
    B = Mzd.random(mb, nb, 600 + mb)
    want = (oracle.trsm_upper_right if upper else oracle.trsm_lower_right)(T, B.copy())
    got = B.copy()
    getattr(m4ri_amd.lib(), "mzd_trsm_upper_right" if upper else "mzd_trsm_lower_right")(T.ptr, got.ptr, 0)
    assert np.array_equal(got.valid_words(), want.valid_words())


def test_trsm_big_blocks_off():
    """The same shapes and the rest of this file in a child process with the 4096-row block inverses switched off: the
    512-row path alone (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, M4RI_AMD_TRSM_BIG="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_trsm.py"), "-x", "-q", "-m", "gpu", "-k", "not blocks_off", "-p", "no:cacheprovider"],
                       cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("mb", [700, 1500])
def test_solves_on_ragged_windows_of_pinned_parents(oracle, mb):
    """T = a window of a PINNED parent whose column count is off the word grid: its last word carries the parent's
    neighbouring columns on the device.  Upper-left, lower-left and both right-hand solves must not see them (the shim solves
    from a masked copy of such a T, mzd_api.hip: run_trsm), nor touch anything outside B's window."""
    n = 2100
    PT, PB, PR = Mzd.random(n, n, 71), Mzd.random(n, n, 72), Mzd.random(n, n, 73)
    HT, HB, HR = PT.copy(), PB.copy(), PR.copy()
    for P in (PT, PB, PR):
        m4ri_amd.pin(P)
    t, h = PT.window(5, 64, 5 + mb, 64 + mb), HT.window(5, 64, 5 + mb, 64 + mb)          # mb x mb, ragged last word
    b, hb = PB.window(11, 128, 11 + mb, 128 + 777), HB.window(11, 128, 11 + mb, 128 + 777)   # mb x 777 (left solves)
    r, hr = PR.window(3, 0, 3 + 500, mb), HR.window(3, 0, 3 + 500, mb)                       # 500 x mb (right solves)
    m4ri_amd.mzd_trsm_upper_left(t, b)
    oracle.trsm_upper_left(h, hb)
    m4ri_amd.mzd_trsm_lower_left(t, b)
    oracle.trsm_lower_left(h, hb)
    m4ri_amd.lib().mzd_trsm_upper_right(t.ptr, r.ptr, 0)
    oracle.trsm_upper_right(h, hr)
    m4ri_amd.lib().mzd_trsm_lower_right(t.ptr, r.ptr, 0)
    oracle.trsm_lower_right(h, hr)
    for P in (PT, PB, PR):
        m4ri_amd.unpin(P)
    assert np.array_equal(PT.rows(), HT.rows()) and np.array_equal(PB.rows(), HB.rows()) and np.array_equal(PR.rows(), HR.rows())
