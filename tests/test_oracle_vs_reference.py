"""Pin the oracle: oracle/gf2_oracle.c must agree bit for bit with the REAL reference (built from
/root/reference into oracle/_ref/ by oracle/Makefile) on every shape of the reference's own
multiply tests, on window cases with non-zero excess, and with an independent numpy product.

CPU only.  Skipped (not failed) where oracle/_ref is absent.
"""
import numpy as np
import pytest

import shapes
from m4ri_amd.mzd import Mzd


def _triple(m, l, n, tag):
    A = Mzd.random(m, l, shapes.seed_of(tag, m, l, n, 1))
    B = Mzd.random(l, n, shapes.seed_of(tag, m, l, n, 2))
    return A, B


@pytest.mark.parametrize("m,l,n,k,cutoff", shapes.MUL + shapes.EDGE)
def test_mul_matches_reference(oracle, reference, m, l, n, k, cutoff):
    A, B = _triple(m, l, n, 11)
    want = reference.mul(None, A, B, cutoff)
    assert oracle.mul(None, A, B, cutoff).equal(want)
    # leaf and definitional product agree too (test_multiplication.c:17-73 mul_test_equality)
    assert oracle.mul_m4rm(Mzd.init(m, n), A, B, k, 1).equal(want)
    assert oracle.mul_naive(Mzd.init(m, n), A, B, 1).equal(want)
    if m and l and n:  # the reference's small-shape fallback aborts on empty operands (mzd_transpose/mzd_copy)
        assert reference.mul_m4rm(None, A, B, k).equal(want)


@pytest.mark.parametrize("m,l,n,k,cutoff", shapes.ADDMUL)
def test_addmul_matches_reference(oracle, reference, m, l, n, k, cutoff):
    A, B = _triple(m, l, n, 12)
    C0 = Mzd.random(m, n, shapes.seed_of(12, m, l, n, 3))
    want = reference.addmul(C0.copy(), A, B, cutoff)
    assert oracle.addmul(C0.copy(), A, B, cutoff).equal(want)
    assert oracle.mul_m4rm(C0.copy(), A, B, k, 0).equal(want)
    assert reference.addmul_m4rm(C0.copy(), A, B, k).equal(want)


@pytest.mark.parametrize("n,k,cutoff", shapes.SQR)
def test_sqr_matches_reference(oracle, reference, n, k, cutoff):
    A = Mzd.random(n, n, shapes.seed_of(13, n))
    want = reference.mul(None, A, A, cutoff)  # A == B: the reference's _mzd_sqr_even path
    assert oracle.mul(None, A, A, cutoff).equal(want)


@pytest.mark.parametrize("n,k,cutoff", shapes.ADDSQR)
def test_addsqr_matches_reference(oracle, reference, n, k, cutoff):
    A = Mzd.random(n, n, shapes.seed_of(14, n))
    C0 = Mzd.random(n, n, shapes.seed_of(14, n, 3))
    want = reference.addmul(C0.copy(), A, A, cutoff)
    assert oracle.addmul(C0.copy(), A, A, cutoff).equal(want)


@pytest.mark.parametrize("M,N,m,n", shapes.SMALLOPS)
def test_windows_with_excess(oracle, reference, M, N, m, n):
    """test_smallops.c:9-108: windows inside a pattern-filled parent; products written into a window
    must leave every parent bit outside the window untouched (mzd_check_pattern, testing.c:22-37)."""
    def parent():
        P = Mzd.init(M, N)
        P.rows()[:, :] = np.uint64(shapes.SMALLOPS_PATTERN)
        return P

    PA, PB, PC_o, PC_r = parent(), parent(), parent(), parent()
    k = min(m, n)
    a, b = PA.window(0, 0, m, k), PB.window(0, 0, k, n)
    a.fill_splitmix(shapes.seed_of(15, M, N, 1))
    b.fill_splitmix(shapes.seed_of(15, M, N, 2))
    co, cr = PC_o.window(0, 0, m, n), PC_r.window(0, 0, m, n)
    co.fill_splitmix(77)
    cr.fill_splitmix(77)
    truth = Mzd.from_bits(((a.to_bits().astype(np.int64) @ b.to_bits().astype(np.int64)) & 1).astype(np.uint8))
    # cutoff 0 = what test_smallops.c runs (leaf only at these sizes): whole parents must be identical,
    # pattern and excess bits included
    oracle.mul(co, a, b, 0)
    reference.mul(cr, a, b, 0)
    assert np.array_equal(PC_o.buf, PC_r.buf) and co.equal(truth)
    oracle.addmul(co, a, b, 0)
    reference.addmul(cr, a, b, 0)
    assert np.array_equal(PC_o.buf, PC_r.buf)
    # cutoff 64 forces Strassen levels + remainder strips ON WINDOWS WITH NON-ZERO EXCESS.  The
    # reference never tests this and gets a valid bit wrong there: its < 54-column fallback
    # (brilliantrussian.c:1063 -> mzd_mul_naive, mzd.c:1141) mis-handles a windowed operand whose
    # excess bits are non-zero (verified: same call is right when the parent pattern is 0).  So the
    # oracle is checked against the arithmetic truth here, and the reference only where it is right.
    oracle.mul(co, a, b, 64)
    assert co.equal(truth)
    reference.mul(cr, a, b, 64)
    if cr.equal(truth):
        assert np.array_equal(PC_o.buf, PC_r.buf)
    # nothing outside the window moved: rows below, words right of it, and the excess bits of its last word
    pat = np.uint64(shapes.SMALLOPS_PATTERN)
    assert np.all(PC_o.rows()[m:, :] == pat) and np.all(PC_o.rows()[:m, co.width:] == pat)
    if n % 64:
        keep = ~np.uint64(co.high_bitmask)
        assert np.all((PC_o.rows()[:m, co.width - 1] & keep) == (pat & keep))
    assert np.all(PA.rows()[m:, :] == pat) and np.all(PB.rows()[k:, :] == pat)


def test_add_matches_reference(oracle, reference):
    for (r, c) in [(1, 1), (10, 64), (33, 65), (100, 511), (64, 1024)]:
        A, B = Mzd.random(r, c, 5), Mzd.random(r, c, 6)
        assert oracle.add(Mzd.init(r, c), A, B).equal(reference.add(Mzd.init(r, c), A, B))


@pytest.mark.parametrize("m,l,n", [(1, 1, 1), (5, 70, 130), (64, 64, 64), (100, 257, 33), (130, 200, 190)])
def test_oracle_vs_numpy(oracle, m, l, n):
    """Independent third opinion: integer matmul mod 2 on unpacked bits."""
    A, B = _triple(m, l, n, 16)
    want = Mzd.from_bits(((A.to_bits().astype(np.int64) @ B.to_bits().astype(np.int64)) & 1).astype(np.uint8))
    for cutoff in (0, 64):
        assert oracle.mul(None, A, B, cutoff).equal(want)


def test_fill_and_fingerprint_agree(oracle):
    """The numpy fill (m4ri_amd.mzd) and the C fill are the same stream in the same order."""
    for (r, c) in [(3, 1), (7, 64), (5, 65), (9, 200)]:
        a = Mzd.random(r, c, 99)
        b = Mzd.init(r, c)
        oracle.fill(b, 99)
        assert np.array_equal(a.buf, b.buf)
        assert a.fingerprint() == oracle.fingerprint(b)


def test_fill_matches_reference_randomize_custom(oracle, reference):
    """Mzd.random / gf2o_fill_splitmix / the device fill all claim the fill order of the reference's
    mzd_randomize_custom (mzd.c:1282-1292: `width` callback words per row, row-major, the last one
    merged under the column mask).  Pin that claim against the reference itself, driven by a splitmix64
    callback -- plain matrices, and windows inside a pattern-filled parent (widths of
    tests/test_random.c:33-62: n + {1, 2, 32, 50, 51, 52, 63, 64, 65})."""
    import ctypes
    from m4ri_amd.mzd import MzdPtr, splitmix_words
    CB = ctypes.CFUNCTYPE(ctypes.c_uint64, ctypes.c_void_p)
    fn = reference.L.mzd_randomize_custom
    fn.restype, fn.argtypes = None, [MzdPtr, CB, ctypes.c_void_p]

    def ref_fill(M, seed):
        state = {"i": 0}

        def next_word(_):
            w = int(splitmix_words(seed, state["i"], 1)[0])
            state["i"] += 1
            return w
        cb = CB(next_word)
        fn(M.ptr, cb, None)

    for base in (0, 64, 128):
        for extra in (1, 2, 32, 50, 51, 52, 63, 64, 65):
            c = base + extra
            a = Mzd.random(7, c, 1234 + c)
            b = Mzd.init(7, c)
            ref_fill(b, 1234 + c)
            assert np.array_equal(a.buf, b.buf), (7, c)
            # window with non-zero excess: the parent's bits outside the window survive
            P1, P2 = Mzd.random(9, c + 130, 77), Mzd.random(9, c + 130, 77)
            w1, w2 = P1.window(1, 64, 8, 64 + c), P2.window(1, 64, 8, 64 + c)
            w1.fill_splitmix(5)
            ref_fill(w2, 5)
            assert np.array_equal(P1.buf, P2.buf), ("window", c)


def test_reference_mul_mp_needs_a_zero_result_block_for_ragged_shapes(oracle):
    """Pins a defect of the reference the parity runs have to know about (found by tests/soak_large.py): mzd_mul_mp(C, A, B) is documented
    as C = AB with a preallocated C (mp.h:34-47), but _mzd_mul_mp4 (mp.c:212-235) ADDS the remainder strips of a ragged product onto
    whatever C held -- so it equals mzd_mul only on a zero result block (or dimensions that are multiples of 128).  libm4ri_amd.so's
    mzd_mul_mp overwrites C like mzd_mul does; a checker that calls the reference's multi-core path must hand it a zero C."""
    import cpu_libs
    omp = cpu_libs.reference(openmp=True)
    if omp is None or not omp.has_mp:
        pytest.skip("oracle/_ref/libm4ri_ref_omp.so not built")
    m, l, n = 6100, 6250, 6190           # above the cutoff in every dimension, none a multiple of 128
    A, B = Mzd.random(m, l, 71), Mzd.random(l, n, 72)
    want = oracle.mul(None, A, B, 0)
    assert omp.mul_mp(Mzd.init(m, n), A, B, 0).equal(want)
    dirty = omp.mul_mp(Mzd.random(m, n, 73), A, B, 0)
    assert not dirty.equal(want)
    a, c = m - m % 128, n - n % 128      # the even block is a product, the strips are not
    assert np.array_equal(dirty.window(0, 0, a, c).masked(), want.window(0, 0, a, c).masked())
    mm, ll, nn = 6144, 6272, 6400        # multiples of 128: no strips, no defect
    A, B = Mzd.random(mm, ll, 74), Mzd.random(ll, nn, 75)
    assert omp.mul_mp(Mzd.random(mm, nn, 76), A, B, 0).equal(oracle.mul(None, A, B, 0))
