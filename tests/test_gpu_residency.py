"""Residency API (include/m4ri_amd.h part 3): pinned matrices and windows into them are used in place
on the device by the M4RI-named entry points; results stay on the device until sync/unpin.  Checked
against the oracle performing the same calls on host copies (the TRSM/PLE pattern of
mzd_addmul on windows of a few matrices, triangular.c:100,348,439,503, ple.c:126)."""
import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)


def test_pinned_product_stays_on_device_until_sync(oracle):
    A, B = Mzd.random(700, 900, 1), Mzd.random(900, 1100, 2)
    C = Mzd.init(700, 1100)
    want = oracle.mul(None, A, B, 0)
    for M in (A, B, C):
        m4ri_amd.pin(M)
        assert m4ri_amd.is_pinned(M) == 1
    m4ri_amd.mzd_mul(C, A, B, 0)
    assert m4ri_amd.is_pinned(C) == 2                 # device copy newer
    assert not C.rows().any()                         # the host copy was not touched
    m4ri_amd.sync(C)
    assert m4ri_amd.is_pinned(C) == 1 and C.equal(want)
    # mixed: pinned operands, host result
    D = Mzd.random(700, 1100, 3)
    m4ri_amd.mzd_mul(D, A, B, 0)
    assert D.equal(want)
    # host result accumulates a pinned operand's product with a host operand
    B2 = Mzd.random(900, 1100, 4)
    want2 = oracle.addmul(want.copy(), A, B2, 0)
    m4ri_amd.mzd_addmul(D, A, B2, 0)
    assert D.equal(want2)
    for M in (A, B, C):
        m4ri_amd.unpin(M)
        assert m4ri_amd.is_pinned(M) == 0
    with pytest.raises(ValueError):
        m4ri_amd.sync(C)


@pytest.mark.parametrize("n,cut,hc", [(2048, 1024, 2048), (2048, 1024, 1990), (1500, 704, 1413)])
def test_schur_complement_chain_on_windows_of_pinned_parents(oracle, n, cut, hc):
    """Block elimination pattern: M22 += M21 * M12, then M12 += M11 * M12' ... all windows of ONE pinned
    parent (hc < n leaves windows whose last word is shared with the parent's other columns)."""
    P = Mzd.random(n, n, 11)
    Q = Mzd.random(n, n, 12)
    Ph, Qh = P.copy(), Q.copy()  # host twins for the oracle

    def blocks(M):
        return (M.window(0, 0, cut, cut), M.window(0, cut, cut, hc), M.window(cut, 0, n, cut), M.window(cut, cut, n, hc))

    m4ri_amd.pin(P)
    m4ri_amd.pin(Q)
    p11, p12, p21, p22 = blocks(P)
    q11, q12, q21, q22 = blocks(Q)
    h11, h12, h21, h22 = blocks(Ph)
    g11, g12, g21, g22 = blocks(Qh)
    # (C, A, B, add) sequences mixing both parents; results feed later steps
    seq_dev = [(p22, p21, p12, True), (q22, p21, q12, True), (p12, q11, q12, False), (q21, p21, q11, True)]
    seq_host = [(h22, h21, h12, True), (g22, h21, g12, True), (h12, g11, g12, False), (g21, h21, g11, True)]
    for (c, a, b, add), (ch, ah, bh, _) in zip(seq_dev, seq_host):
        if add:
            m4ri_amd.mzd_addmul(c, a, b, 0)
            oracle.addmul(ch, ah, bh, 0)
        else:
            m4ri_amd.mzd_mul(c, a, b, 0)
            oracle.mul(ch, ah, bh, 0)
    assert m4ri_amd.is_pinned(P) == 2 and m4ri_amd.is_pinned(Q) == 2
    m4ri_amd.unpin(P)
    m4ri_amd.unpin(Q)
    assert np.array_equal(P.buf, Ph.buf), "parent P (every word, incl. columns outside the windows)"
    assert np.array_equal(Q.buf, Qh.buf), "parent Q"


def test_host_modified_refreshes_the_device_copy(oracle):
    A, B = Mzd.random(300, 320, 21), Mzd.random(320, 200, 22)
    m4ri_amd.pin(A)
    A.rows()[:, :] = Mzd.random(300, 320, 23).rows()      # host overwrites A behind the device's back
    m4ri_amd.host_modified(A)
    got = m4ri_amd.mzd_mul(None, A, B, 0)
    m4ri_amd.unpin(A)
    assert got.equal(oracle.mul(None, A, B, 0))


def test_pin_rejects_windows():
    A = Mzd.random(128, 128, 31)
    with pytest.raises(ValueError):
        m4ri_amd.pin(A.window(0, 0, 64, 64))


def test_randomized_chain_on_pinned_parents(oracle):
    """Seeded fuzz: 60 products whose operands and results are random windows of three pinned parents
    (results feed later products; the result's parent is never an operand's parent), mirrored by the
    oracle on host twins; every word of every parent must agree at the end."""
    rng = np.random.default_rng(7)
    dims = [(1500, 1700), (1300, 1500), (1600, 1400)]
    dev = [Mzd.random(r, c, 40 + i) for i, (r, c) in enumerate(dims)]
    host = [M.copy() for M in dev]
    for M in dev:
        m4ri_amd.pin(M)

    def win(idx, rows, cols):
        pr, pc = dims[idx]
        lowr = int(rng.integers(0, pr - rows + 1))
        lowc = int(rng.integers(0, (pc - cols) // 64 + 1)) * 64
        return dev[idx].window(lowr, lowc, lowr + rows, lowc + cols), host[idx].window(lowr, lowc, lowr + rows, lowc + cols)

    for case in range(60):
        ci = int(rng.integers(0, 3))
        ai, bi = ((ci + 1) % 3, (ci + 2) % 3) if rng.integers(0, 2) else ((ci + 2) % 3, (ci + 2) % 3)
        m, l, n = int(rng.integers(1, 1200)), int(rng.integers(1, 1200)), int(rng.integers(1, 1200))
        (A, Ah), (B, Bh), (C, Ch) = win(ai, m, l), win(bi, l, n), win(ci, m, n)
        cutoff = int(rng.choice([0, 64, 256]))
        if rng.integers(0, 2):
            m4ri_amd.mzd_addmul(C, A, B, cutoff)
            oracle.addmul(Ch, Ah, Bh, cutoff)
        else:
            m4ri_amd.mzd_mul(C, A, B, cutoff)
            oracle.mul(Ch, Ah, Bh, cutoff)
    for M, H in zip(dev, host):
        m4ri_amd.unpin(M)
        assert np.array_equal(M.buf, H.buf)


def test_full_width_ragged_pinned_result_keeps_zero_excess(oracle):
    """B is a ragged window of a WIDER pinned parent (its last word carries the neighbouring columns), C a
    pinned non-window matrix with ncols % 64 != 0: the engine writes C in place, and the excess bits of a
    non-window matrix must be zero afterwards (mzd.h:115-121) -- on the device copy and after sync."""
    m, l, n, wide = 300, 257, 131, 400
    A = Mzd.random(m, l, 21)
    PB = Mzd.random(l, wide, 22)
    Bw = PB.window(0, 0, l, n)                # columns 131..399 of the parent are B's "excess"
    C = Mzd.init(m, n)
    want = oracle.mul(None, A, Bw.copy(), 0)
    for M in (A, PB, C):
        m4ri_amd.pin(M)
    m4ri_amd.mzd_mul(C, A, Bw, 0)
    m4ri_amd.sync(C)
    assert C.equal(want)
    assert not (C.valid_words()[:, -1] & ~np.uint64(C.high_bitmask)).any(), "excess bits of a non-window result must be 0"
    # accumulate through the same path, then a host-side word-wise consumer sees clean words
    want2 = oracle.addmul(want.copy(), A, Bw.copy(), 0)
    m4ri_amd.mzd_addmul(C, A, Bw, 0)
    m4ri_amd.unpin(C)
    assert np.array_equal(C.valid_words(), want2.masked())
    m4ri_amd.unpin(A)
    m4ri_amd.unpin(PB)


def test_repinning_a_reused_address_uploads_again(oracle):
    """A matrix dropped without unpin leaves a registry entry; a new matrix at the same address must not
    be served from the dead device copy (m4ri_amd_pin detects the stale entry)."""
    A = Mzd.random(200, 300, 31)
    B = Mzd.random(300, 260, 32)
    m4ri_amd.pin(A)
    A2 = Mzd(200, 300, buf=A.buf)             # same host block, another descriptor ("freed and reallocated")
    A2.fill_splitmix(33)
    m4ri_amd.pin(A2)
    want = oracle.mul(None, A2, B, 0)
    assert m4ri_amd.mzd_mul(None, A2, B, 0).equal(want)
    m4ri_amd.unpin(A2)
