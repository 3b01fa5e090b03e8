"""Products below the GPU's crossover: the library's own host Method of Four Russians (m4ri_amd/csrc/small_host.cpp; the
counterpart of the reference's switch to mzd_mul_naive inside _mzd_mul_m4rm, m4ri/brilliantrussian.c:1063-1068,
m4ri/mzd.c:1141-1172).  The routine itself against the oracle on the reference's own shape lists (no GPU needed); on the GPU box,
that the entry points take it exactly when they should and never when an operand lives on the device."""
import numpy as np
import pytest

import m4ri_amd
import shapes
from m4ri_amd.mzd import Mzd


@pytest.mark.parametrize("m,l,n,k,cutoff", shapes.MUL + shapes.EDGE)
def test_host_routine_mul_vs_oracle(oracle, m, l, n, k, cutoff):
    A, B = Mzd.random(m, l, shapes.seed_of(m, l, n, 1)), Mzd.random(l, n, shapes.seed_of(m, l, n, 2))
    C = Mzd.random(m, n, 5)                       # a dirty C is overwritten, its excess bits end up zero
    m4ri_amd.small_mul_host(C, A, B, False)
    assert C.equal(oracle.mul(None, A, B, 0)) and not (C.rows()[:, C.width:] != 0).any()
    if m and n and n % 64:
        assert not (C.valid_words()[:, -1] & ~np.uint64(C.high_bitmask)).any()


@pytest.mark.parametrize("m,l,n,k,cutoff", shapes.ADDMUL + [(5, 0, 7, 0, 0)])
def test_host_routine_addmul_vs_oracle(oracle, m, l, n, k, cutoff):
    A, B, C = Mzd.random(m, l, 11), Mzd.random(l, n, 12), Mzd.random(m, n, 13)
    want = oracle.addmul(C.copy(), A, B, 0) if l else C.copy()
    assert m4ri_amd.small_mul_host(C, A, B, True).equal(want)


def test_host_routine_squares_and_empty_inner(oracle):
    for n in (1, 64, 131, 193, 300):
        A = Mzd.random(n, n, 21 + n)
        assert m4ri_amd.small_mul_host(Mzd.init(n, n), A, A, False).equal(oracle.mul(None, A, A, 0))
    for (m, l, n) in shapes.EMPTY_INNER:
        C = Mzd.random(m, n, 3)
        m4ri_amd.small_mul_host(C, Mzd.init(m, l), Mzd.init(l, n), False)
        assert not C.valid_words().any()


@pytest.mark.parametrize("M,N,m,n", shapes.SMALLOPS)
def test_host_routine_windows_keep_their_parents(oracle, M, N, m, n):
    """Operands and result are windows with dirty bits around them (tests/test_smallops.c:115-121): every bit of C's parent outside
    the window survives, the bits of A's and B's parents outside their windows do not leak in."""
    PA, PB, PC = Mzd.random(M, N, 31), Mzd.random(M, N, 32), Mzd.random(M, N, 33)
    lowc = 64 if N - 64 >= max(m, n) else 0
    a, b = PA.window(0, 0, m, n), PB.window(1 if M > n else 0, lowc, (1 if M > n else 0) + n, lowc + m)
    c = PC.window(M - m, 0, M, m)
    want_parent = Mzd(M, N, buf=PC.buf.copy())
    co = want_parent.window(M - m, 0, M, m)
    oracle.mul(co, a.copy(), b.copy(), 0)
    m4ri_amd.small_mul_host(c, a, b, False)
    assert np.array_equal(PC.buf, want_parent.buf)
    oracle.addmul(co, a.copy(), b.copy(), 0)
    m4ri_amd.small_mul_host(c, a, b, True)
    assert np.array_equal(PC.buf, want_parent.buf)


def test_threshold_is_a_plain_setting():
    old = m4ri_amd.set_small_product_threshold(12345)
    assert m4ri_amd.set_small_product_threshold(-1) == 12345 and m4ri_amd.set_small_product_threshold(old) == 12345


@pytest.mark.gpu
def test_entry_points_take_the_host_routine_exactly_when_they_should(oracle):
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1
    m4ri_amd.init(0)
    old = m4ri_amd.set_small_product_threshold(1 << 24)  # (the test pins its own threshold)
    try:
        for (m, l, n) in [(1, 1, 1), (64, 64, 64), (21, 171, 31), (193, 65, 65), (256, 256, 256), (1000, 10, 20)]:
            A, B, C0 = Mzd.random(m, l, 41), Mzd.random(l, n, 42), Mzd.random(m, n, 43)
            before = m4ri_amd.small_product_count()
            assert m4ri_amd.mzd_mul(None, A, B, 0).equal(oracle.mul(None, A, B, 0))
            assert m4ri_amd.mzd_addmul(C0.copy(), A, B, 0).equal(oracle.addmul(C0.copy(), A, B, 0))
            assert m4ri_amd.mzd_mul_m4rm(None, A, B, 0).equal(oracle.mul(None, A, B, 0))
            took = m4ri_amd.lib().m4ri_amd_small_product_wanted(m, l, n)   # 256^3 is m * l * n == 2^24 but over the routine's own cost bound: the GPU
            assert took == (0 if (m, l, n) == (256, 256, 256) else 1) and m4ri_amd.small_product_count() == before + 3 * took, (m, l, n)
        A, B = Mzd.random(300, 300, 44), Mzd.random(300, 300, 45)        # 2.7e7 > 2^24: the GPU
        before = m4ri_amd.small_product_count()
        assert m4ri_amd.mzd_mul(None, A, B, 0).equal(oracle.mul(None, A, B, 0)) and m4ri_amd.small_product_count() == before
        A, B, C = Mzd.random(128, 128, 46), Mzd.random(128, 128, 47), Mzd.init(128, 128)
        for M in (A, B, C):                                               # operands on the device: the product goes where they are
            m4ri_amd.pin(M)
        m4ri_amd.mzd_mul(C, A, B, 0)
        assert m4ri_amd.small_product_count() == before and m4ri_amd.is_pinned(C) == 2
        for M in (A, B, C):
            m4ri_amd.unpin(M)
        assert C.equal(oracle.mul(None, A, B, 0))
        m4ri_amd.set_small_product_threshold(0)
        A, B = Mzd.random(64, 64, 48), Mzd.random(64, 64, 49)
        assert m4ri_amd.mzd_mul(None, A, B, 0).equal(oracle.mul(None, A, B, 0)) and m4ri_amd.small_product_count() == before
    finally:
        m4ri_amd.set_small_product_threshold(old)


def test_host_routine_fuzz_shapes_and_windows(oracle):
    """Seeded fuzz of the host routine alone: random shapes around its regimes (at most 8 rows: no tables; 3- to 8-bit tables by the
    number of rows; one to several column blocks), mul and addmul, operands and result windows of larger parents; every word of C's
    parent is compared."""
    import os
    rng = np.random.default_rng(int(os.environ.get("M4RI_AMD_FUZZ_SEED", "20260929")))
    for case in range(int(os.environ.get("M4RI_AMD_FUZZ_CASES", "150"))):
        m = int(rng.choice([rng.integers(1, 16), rng.integers(16, 224), rng.integers(224, 400), rng.integers(400, 1600)]))
        l, n = int(rng.integers(1, 400)), int(rng.choice([rng.integers(1, 400), rng.integers(400, 1300)]))
        add = bool(rng.integers(0, 2))

        def operand(rows, cols, seed):
            if rng.integers(0, 2):
                return Mzd.random(rows, cols, seed), None, (0, 0)
            pr, pc = rows + int(rng.integers(0, 9)), cols + int(rng.integers(0, 200))
            parent = Mzd.random(pr, pc, seed)
            lowr, lowc = int(rng.integers(0, pr - rows + 1)), int(rng.integers(0, (pc - cols) // 64 + 1)) * 64
            return parent.window(lowr, lowc, lowr + rows, lowc + cols), parent, (lowr, lowc)
        A, _, _ = operand(m, l, 1000 + case)
        B, _, _ = operand(l, n, 2000 + case)
        C, Cp, (r0, c0) = operand(m, n, 3000 + case)
        if Cp is None:
            want = oracle.addmul(C.copy(), A.copy(), B.copy(), 0) if add else oracle.mul(None, A.copy(), B.copy(), 0)
            m4ri_amd.small_mul_host(C, A, B, add)
            assert C.equal(want), (case, m, l, n, add)
        else:
            wp = Mzd(Cp.nrows, Cp.ncols, buf=Cp.buf.copy())
            wc = wp.window(r0, c0, r0 + m, c0 + n)
            (oracle.addmul if add else oracle.mul)(wc, A.copy(), B.copy(), 0)
            m4ri_amd.small_mul_host(C, A, B, add)
            assert np.array_equal(Cp.buf, wp.buf), (case, m, l, n, add)


def test_which_products_the_host_routine_takes_is_a_cost_rule():
    """m * l * n alone sent 1 x 1 x 2^26 and 2^26 x 1 x 1 to a single-threaded loop: the rule also bounds the
    routine's own cost in word operations of ITS algorithm (pure arithmetic: m4ri_amd_small_product_wanted, default threshold 2^27 since
    the routine's second generation: profiles/r06_small_products_host_routine.log)."""
    old = m4ri_amd.set_small_product_threshold(1 << 27)
    try:
        w = m4ri_amd.lib().m4ri_amd_small_product_wanted
        for shape in [(1, 1, 1), (64, 64, 64), (256, 256, 256), (384, 384, 384), (448, 448, 448), (512, 512, 512), (1000, 10, 20), (16, 4096, 16), (4096, 16, 64),
                      (64, 64, 4096), (2048, 64, 64), (1024, 256, 256), (256, 1024, 256), (256, 256, 1024), (2048, 16, 2048), (512, 512, 8)]:
            assert w(*shape) == 1, shape                 # the measured wins against the GPU path stay on the host
        for shape in [(1, 1, 1 << 26), (1 << 26, 1, 1), (1, 1 << 26, 1), (8, 8192, 1024), (576, 576, 576), (1024, 1024, 1024), (1 << 20, 8, 8),
                      (100000, 1, 600), (2048, 2048, 16), (4096, 64, 4096), (128, 128, 8192)]:
            assert w(*shape) == 0, shape                 # degenerate, or the GPU path is as fast or faster: the GPU
        assert w(0, 5, 5) == 0 and w(5, 0, 5) == 1       # an empty inner dimension is a clear of C: nothing to upload
        m4ri_amd.set_small_product_threshold(1 << 26)    # the bound scales with the threshold
        assert w(384, 384, 384) == 1 and w(448, 448, 448) == 0
        m4ri_amd.set_small_product_threshold(0)
        assert w(4, 4, 4) == 0
    finally:
        m4ri_amd.set_small_product_threshold(old)


@pytest.mark.gpu
def test_parity_at_the_default_threshold(oracle):
    """One parity run with the routing a program under LD_PRELOAD gets (the session fixture forces every other test's products onto the
    GPU): the reference's own shape lists through the entry points at the DEFAULT threshold, each product on the side the rule
    names, every result the oracle's."""
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1
    m4ri_amd.init(0)
    old = m4ri_amd.set_small_product_threshold(1 << 27)
    try:
        took_host = took_gpu = 0
        for (m, l, n, k, cutoff) in shapes.MUL + shapes.EDGE + [(1, 1, 70000, 0, 0), (3000, 2, 3000, 0, 0)]:
            A, B = Mzd.random(m, l, shapes.seed_of(m, l, n, 1)), Mzd.random(l, n, shapes.seed_of(m, l, n, 2))
            before = m4ri_amd.small_product_count()
            got = m4ri_amd.mzd_mul(None, A, B, cutoff)
            assert got.equal(oracle.mul(None, A, B, 0)), (m, l, n)
            host = m4ri_amd.small_product_count() - before
            if m and n and l:
                assert host == m4ri_amd.lib().m4ri_amd_small_product_wanted(m, l, n), (m, l, n, host)
            took_host += host
            took_gpu += 1 - host
        assert took_host > 5 and took_gpu > 5, (took_host, took_gpu)
    finally:
        m4ri_amd.set_small_product_threshold(old)
