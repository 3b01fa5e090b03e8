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


@pytest.mark.parametrize("m,n", [(1, 70), (70, 1), (13, 17), (64, 64), (65, 63), (127, 129), (200, 70), (70, 200), (300, 300), (513, 511), (1000, 200),
                                 (200, 1000), (1025, 1025)])
@pytest.mark.parametrize("kind", ["random", "lowrank", "sparse", "zerocols"])
def test_pluq_matches_reference(oracle, reference, m, n, kind):
    """PLE + the column step of PLUQ (ple.c:50-60, mzp.c:279-293) against mzd_pluq and _mzd_pluq_russian."""
    A = _make(kind, m, n, 2000 + 7 * m + n)
    Ao, Ar, Ar2 = A.copy(), A.copy(), A.copy()
    want = oracle.ple(Ao, pluq=True)
    for Ax, which in ((Ar, "mzd_pluq"), (Ar2, "_mzd_pluq_russian")):
        got = reference.ple(Ax, which)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), which
        assert np.array_equal(Ax.valid_words(), Ao.valid_words()), which


def _defects(m, n, seed, dup_rows=0, zero_rows=0):
    """Random matrix with columns that repeat their left neighbour (so no pivot there) scattered over the whole width, a
    zero word column, optionally copies of the first rows at the bottom (they vanish during the elimination: windows
    of the recursion end in zero rows) and zero rows below those."""
    A = Mzd.random(m, n, seed)
    w = A.valid_words()
    rng = np.random.default_rng(seed)
    for c in sorted(rng.choice(np.arange(1, n), size=max(3, n // 40), replace=False)):
        src, dst = int(c) - 1, int(c)
        bit = (w[:, src // 64] >> np.uint64(src % 64)) & np.uint64(1)
        w[:, dst // 64] = (w[:, dst // 64] & ~(np.uint64(1) << np.uint64(dst % 64))) | (bit << np.uint64(dst % 64))
    w[:, (n // 3) // 64] = 0
    if dup_rows:
        w[m - zero_rows - dup_rows: m - zero_rows] = w[:dup_rows]
    if zero_rows:
        w[m - zero_rows:] = 0
    return A


RECURSIVE_CASES = [(4200, 8256, 0, 0), (9000, 4200, 0, 0), (6000, 11000, 0, 0), (8400, 4800, 700, 300), (5000, 9000, 400, 0), (3000, 12000, 0, 500)]


@pytest.mark.parametrize("m,n,dup,zero", RECURSIVE_CASES)
def test_recursive_ple_and_pluq_match_reference(oracle, reference, m, n, dup, zero):
    """Above __M4RI_PLE_CUTOFF (width * nrows > 524288 words) mzd_ple / mzd_pluq recurse on column halves (ple.c:62-171) and
    leave their own values in Q behind the rank; mzd_pluq then applies those transpositions too.  The oracle's
    restatement of that recursion against the reference: matrix, P, the whole of Q, rank."""
    A = _defects(m, n, 3000 + m + n, dup, zero)
    for pluq, which in ((False, "mzd_ple"), (True, "mzd_pluq")):
        Ao, Ar = A.copy(), A.copy()
        want = reference.ple(Ar, which)
        got = oracle.ple(Ao, pluq=pluq, recursive=True)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), which
        assert np.array_equal(got[2][: got[0]], want[2][: got[0]]), which + ": pivots"
        assert np.array_equal(got[2], want[2]), which + ": Q behind the rank"
        assert np.array_equal(Ar.valid_words(), Ao.valid_words()), which
    if n < 12000:  # (the last case has all its pivots in the left half: there it is the early return on a zero window that matters)
        flat = oracle.ple(A.copy())
        assert flat[0] == want[0] and not np.array_equal(flat[2], want[2]), "the case does not exercise the leftovers of the recursion"
