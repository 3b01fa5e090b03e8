"""The oracle's PLE (column by column) pinned against the reference's _mzd_ple_russian
(m4ri/ple_russian.c:380-617: blocks of up to 56 columns, seven Gray-code tables, lazy row updates): the
decomposed matrix, both permutations and the rank, bit for bit -- on full-rank, rank-deficient, tall, wide,
sparse and structured inputs (shapes of the reference's tests/test_ple.c / test_pluq.c, m, n in 1..2 * 64 +- 1
and a few larger ones), and for every table width k."""
import numpy as np
import pytest

from m4ri_amd.mzd import Mzd

SHAPES = [(1, 1), (1, 70), (70, 1), (2, 2), (13, 17), (64, 64), (65, 63), (63, 65), (100, 100), (127, 129), (128, 128), (200, 70),
          (70, 200), (300, 300), (513, 511), (1000, 200), (200, 1000), (1025, 1025), (3000, 512), (2048, 448)]


def _make(kind, m, n, seed):
    A = Mzd.random(m, n, seed)
    if kind == "lowrank" and m > 2 and n > 2:         # rank <= min(m, n) / 3: many columns without a pivot
        r = max(1, min(m, n) // 3)
        X, Y = Mzd.random(m, r, seed + 1).to_bits().astype(np.int64), Mzd.random(r, n, seed + 2).to_bits().astype(np.int64)
        A = Mzd.from_bits(((X @ Y) & 1).astype(np.uint8))
    elif kind == "sparse":                            # ~3 % density: pivots far down, empty column blocks
        b = A.to_bits() & Mzd.random(m, n, seed + 3).to_bits() & Mzd.random(m, n, seed + 4).to_bits()
        b &= Mzd.random(m, n, seed + 5).to_bits() & Mzd.random(m, n, seed + 6).to_bits()
        A = Mzd.from_bits(b)
    elif kind == "zerocols" and n > 70:               # whole 64-column blocks of zeros, then data
        b = A.to_bits()
        b[:, : 64 + 7] = 0
        b[:, n // 2: n // 2 + 3] = 0
        A = Mzd.from_bits(b)
    return A


@pytest.mark.parametrize("m,n", SHAPES)
@pytest.mark.parametrize("kind", ["random", "lowrank", "sparse", "zerocols"])
def test_ple_matches_reference(oracle, reference, m, n, kind):
    A = _make(kind, m, n, 1000 + 7 * m + n)
    Ao, Ar = A.copy(), A.copy()
    ro, Po, Qo = oracle.ple(Ao)
    rr, Pr, Qr = reference.ple(Ar)
    assert ro == rr
    assert np.array_equal(Po, Pr) and np.array_equal(Qo, Qr)
    assert np.array_equal(Ao.valid_words(), Ar.valid_words())


def test_ple_is_independent_of_k(oracle, reference):
    A = Mzd.random(700, 600, 5)
    Ao = A.copy()
    want = oracle.ple(Ao)
    for k in (2, 3, 5, 8):
        Ar = A.copy()
        got = reference.ple(Ar, k=k)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and Ar.equal(Ao)
