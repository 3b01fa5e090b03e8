"""mzd_transpose, mzd_trtri_upper and mzd_trtri_upper_russian on the GPU (include/m4ri_amd.h; reference m4ri/mzd.c:1118-1139,
m4ri/triangular.c:518-547, m4ri/triangular_russian.c:378-470) against the oracle's restatements, which
tests/test_transpose_trtri_oracle.py pins to the reference: every output bit, the untouched bits included."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import m4ri_amd
from m4ri_amd.mzd import Mzd
from test_transpose_trtri_oracle import TRANSPOSE_SIZES, strict_upper_mask, unit_upper

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)
    torch.cuda.set_device(0)


@pytest.mark.parametrize("m", TRANSPOSE_SIZES + [3000, 4000, 5000])  # tests/test_transpose.c:25
def test_transpose_matches_oracle(oracle, m):
    for n in TRANSPOSE_SIZES + [3000, 4000, 5000]:
        if m * n > 3000000 and (m, n) not in ((5000, 5000), (4000, 3000), (1000, 5000), (5000, 1000)):
            continue
        A = Mzd.random(m, n, 100 * m + n)
        want = oracle.transpose(A) if m * n <= 3000000 else Mzd.from_bits(A.to_bits().T.copy())
        got = m4ri_amd.mzd_transpose(A)
        assert (got.nrows, got.ncols) == (n, m)
        assert np.array_equal(got.valid_words(), want.valid_words()), (m, n)
        D = Mzd.random(n, m, 3)  # into a matrix holding other bits (test_transpose.c:44-46)
        m4ri_amd.mzd_transpose(A, D)
        assert np.array_equal(D.valid_words(), want.valid_words())


def test_transpose_windows_keep_their_parents(oracle):
    PA, PD = Mzd.random(300, 900, 1), Mzd.random(700, 500, 2)
    for (r0, c0, m, n), (dr, dc) in (((10, 64, 200, 333), (5, 128)), ((0, 128, 64, 65), (100, 0)), ((7, 0, 293, 650), (20, 64))):
        A = PA.window(r0, c0, r0 + m, c0 + n)
        Po, Pg = PD.copy(), PD.copy()
        oracle.transpose(A, Po.window(dr, dc, dr + n, dc + m))
        m4ri_amd.mzd_transpose(A, Pg.window(dr, dc, dr + n, dc + m))
        assert np.array_equal(Pg.rows(), Po.rows()), "the WHOLE parent of DST must match"


def test_transpose_in_place_and_pinned(oracle):
    A = Mzd.random(640, 640, 4)
    want = oracle.transpose(A)
    G = A.copy()
    m4ri_amd.mzd_transpose(G, G)  # the reference forbids DST == A; staging makes it harmless here
    assert np.array_equal(G.valid_words(), want.valid_words())
    P, D = Mzd.random(500, 700, 5), Mzd(700, 500)
    m4ri_amd.pin(P); m4ri_amd.pin(D)
    try:
        m4ri_amd.mzd_transpose(P, D)
        assert m4ri_amd.is_pinned(D) == 2
        m4ri_amd.sync(D)
        assert np.array_equal(D.valid_words(), oracle.transpose(P).valid_words())
        # a window of a pinned parent whose last word has neighbours
        W = P.window(3, 64, 303, 64 + 130)
        got = m4ri_amd.mzd_transpose(W)
        assert np.array_equal(got.valid_words(), oracle.transpose(W).valid_words())
    finally:
        m4ri_amd.unpin(P); m4ri_amd.unpin(D)


@pytest.mark.parametrize("m,n,stride_pad", [(1, 1, 0), (70, 130, 1), (1024, 1024, 0), (1025, 1023, 1), (2000, 77, 3), (77, 2000, 2), (3000, 2111, 1)])
def test_transpose_dev_strides(m, n, stride_pad):
    """The device entry with odd / padded strides (8-byte store path) and garbage behind A's columns."""
    A = Mzd.random(m, n, 11)
    wa, wd = (n + 63) // 64, (m + 63) // 64
    sa, sd = wa + stride_pad, wd + stride_pad
    host = np.full((m, sa), 0xDEADBEEFDEADBEEF, dtype=np.uint64)
    host[:, :wa] = A.valid_words()
    if n % 64:
        host[:, wa - 1] |= np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(n % 64)  # bits beyond ncols must be ignored
    tA = torch.from_numpy(host.view(np.int64)).cuda()
    tD = torch.full((n, sd), 0x5555555555555555, dtype=torch.int64, device="cuda")
    assert m4ri_amd.lib().m4ri_amd_transpose_dev(tD.data_ptr(), sd, tA.data_ptr(), sa, m, n, None) == 0
    torch.cuda.synchronize()
    got = tD.cpu().numpy().view(np.uint64)
    want = Mzd.from_bits(A.to_bits().T.copy())
    assert np.array_equal(got[:, :wd], want.valid_words())
    assert np.all(got[:, wd:] == np.uint64(0x5555555555555555)), "words beyond D's width are not ours"


def test_transpose_at_scale_properties():
    """65536 x 32768 (test_transpose.c:56-62 at BASELINE sizes): (A^T)^T = A and (A + B)^T = A^T + B^T on the device results."""
    A, B = Mzd.random(65536, 32768, 1), Mzd.random(65536, 32768, 2)
    AT = m4ri_amd.mzd_transpose(A)
    assert np.array_equal(m4ri_amd.mzd_transpose(AT).valid_words(), A.valid_words())
    C = Mzd(65536, 32768)
    C.valid_words()[:] = A.valid_words() ^ B.valid_words()
    BT = m4ri_amd.mzd_transpose(B)
    assert np.array_equal(m4ri_amd.mzd_transpose(C).valid_words(), AT.valid_words() ^ BT.valid_words())
    # a few thousand single bits against the definition
    rng = np.random.default_rng(0)
    ii, jj = rng.integers(0, 65536, 4000), rng.integers(0, 32768, 4000)
    a = (A.valid_words()[ii, jj // 64] >> (jj % 64).astype(np.uint64)) & np.uint64(1)
    t = (AT.valid_words()[jj, ii // 64] >> (ii % 64).astype(np.uint64)) & np.uint64(1)
    assert np.array_equal(a, t)


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 65, 100, 128, 200, 300, 511, 512, 513, 777, 1024, 1100, 1536, 2000, 2049, 3000])
def test_trtri_upper_matches_oracle(oracle, n):
    for keep_lower in (False, True):
        U = unit_upper(n, 40 + n, keep_lower)
        want = oracle.trtri_upper(U.copy())
        for which in ("mzd_trtri_upper", "mzd_trtri_upper_russian"):
            got = m4ri_amd.mzd_trtri_upper(U.copy(), which)
            assert np.array_equal(got.valid_words(), want.valid_words()), (which, keep_lower)


def test_trtri_upper_on_a_window_and_pinned(oracle):
    P = Mzd.random(900, 1000, 6)
    for r0, c0, n in ((10, 64, 700), (100, 128, 513), (0, 0, 900)):
        Po, Pg = P.copy(), P.copy()
        oracle.trtri_upper(Po.window(r0, c0, r0 + n, c0 + n))
        m4ri_amd.mzd_trtri_upper(Pg.window(r0, c0, r0 + n, c0 + n))
        assert np.array_equal(Pg.rows(), Po.rows()), "the WHOLE parent must match"
    U = unit_upper(1300, 7, True)
    want = oracle.trtri_upper(U.copy())
    m4ri_amd.pin(U)
    try:
        m4ri_amd.mzd_trtri_upper(U)
        m4ri_amd.sync(U)
        assert np.array_equal(U.valid_words(), want.valid_words())
    finally:
        m4ri_amd.unpin(U)


@pytest.mark.parametrize("n", [8192 + 192, 16384, 20000])
def test_trtri_upper_at_scale_is_the_inverse(n):
    """tests/test_invert.c:26-33 at sizes the oracle does not reach: U * U^-1 = 1 through the device product, and the
    bits that are not ours untouched."""
    U = unit_upper(n, n, True)
    X = m4ri_amd.mzd_trtri_upper(U.copy())
    mask = strict_upper_mask(n)
    assert np.array_equal(X.valid_words() & ~mask, U.valid_words() & ~mask), "diagonal and lower triangle are the caller's"
    Uc, Xc = Mzd(n, n), Mzd(n, n)
    idx = np.arange(n)
    for dst, src in ((Uc, U), (Xc, X)):
        dst.valid_words()[:] = src.valid_words() & mask
        dst.valid_words()[idx, idx // 64] |= np.uint64(1) << (idx % 64).astype(np.uint64)
    want = Mzd(n, n)
    want.valid_words()[idx, idx // 64] = np.uint64(1) << (idx % 64).astype(np.uint64)
    assert np.array_equal(m4ri_amd.mzd_mul(None, Uc, Xc).valid_words(), want.valid_words())


def test_trtri_and_transpose_at_scale_vs_reference_sha256():
    """Against SHA-256 values of the real reference's results (tests/golden/trtri_transpose.json, make_golden.py --trtri)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trtri_transpose.json")
    assert os.path.exists(path), "committed fixture tests/golden/trtri_transpose.json is missing"
    for e in json.load(open(path)):
        if e["op"] == "trtri":
            X = m4ri_amd.mzd_trtri_upper(unit_upper(e["n"], e["seed"], True))
        else:
            X = m4ri_amd.mzd_transpose(Mzd.random(e["m"], e["n"], e["seed"]))
        assert hashlib.sha256(X.masked().tobytes()).hexdigest() == e["sha256"], e
