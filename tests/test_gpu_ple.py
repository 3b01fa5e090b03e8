"""PLE decomposition on the GPU (include/m4ri_amd.h: mzd_ple / _mzd_ple / _mzd_ple_russian; reference
m4ri/ple.c:33-171, m4ri/ple_russian.c:380-617) against the oracle's column-by-column PLE, which
tests/test_ple_oracle.py pins to the reference: decomposed matrix, P, Q and rank, bit for bit."""
import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd
from test_ple_oracle import SHAPES, _make

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)


def _same(got, want, Ag, Ao):
    assert got[0] == want[0], ("rank", got[0], want[0])
    assert np.array_equal(got[1], want[1]), "P differs"
    assert np.array_equal(got[2], want[2]), "Q differs"
    assert np.array_equal(Ag.valid_words(), Ao.valid_words()), "decomposed matrix differs"


@pytest.mark.parametrize("m,n", SHAPES)
@pytest.mark.parametrize("kind", ["random", "lowrank", "sparse", "zerocols"])
def test_ple_matches_oracle(oracle, m, n, kind):
    A = _make(kind, m, n, 1000 + 7 * m + n)
    Ao = A.copy()
    want = oracle.ple(Ao)
    for which in ("mzd_ple", "_mzd_ple", "_mzd_ple_russian"):
        Ag = A.copy()
        _same(m4ri_amd.mzd_ple(Ag, 0, which), want, Ag, Ao)


@pytest.mark.parametrize("m,n,kind", [(5000, 5000, "random"), (9000, 3000, "random"), (3000, 9000, "random"), (6000, 6000, "lowrank"),
                                     (4099, 8200, "zerocols"), (20000, 512, "random"), (70000, 448, "sparse")])
def test_larger_ple_matches_oracle(oracle, m, n, kind):
    if kind == "lowrank":
        X, Y = Mzd.random(m, 1500, 5), Mzd.random(1500, n, 6)
        A = m4ri_amd.mzd_mul(None, X, Y, 0)
    else:
        A = _make(kind, m, n, 77)
    Ao, Ag = A.copy(), A.copy()
    want = oracle.ple(Ao)
    _same(m4ri_amd.mzd_ple(Ag), want, Ag, Ao)


def test_ple_on_a_window_keeps_the_parent(oracle):
    P0 = Mzd.random(900, 1000, 9)
    for (r0, c0, m, n) in [(10, 64, 500, 333), (0, 0, 900, 130), (100, 128, 64, 64)]:
        Po, Pg = Mzd(900, 1000, buf=P0.buf.copy()), Mzd(900, 1000, buf=P0.buf.copy())
        wo, wg = Po.window(r0, c0, r0 + m, c0 + n), Pg.window(r0, c0, r0 + m, c0 + n)
        want = oracle.ple(wo)
        got = m4ri_amd.mzd_ple(wg)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        assert np.array_equal(Po.buf, Pg.buf)


def test_ple_reconstructs(oracle):
    """A == P * L * E * Q^T-ish identity through independent pieces: rebuild A from the GPU's decomposition with
    numpy (the reference's own check in tests/test_ple.c:26-60 does the same with mzd_mul)."""
    m, n = 300, 420
    A = Mzd.random(m, n, 31)
    D = A.copy()
    r, P, Q = m4ri_amd.mzd_ple(D)
    d = D.to_bits().astype(np.int64)
    L = np.zeros((m, m), dtype=np.int64)
    L[:, :r] = np.tril(d[:, :r], -1)
    L[np.arange(m), np.arange(m)] = 1
    E = np.zeros((m, n), dtype=np.int64)
    for i in range(r):
        E[i, Q[i]:] = d[i, Q[i]:]
        E[i, Q[i]] = 1
    LE = (L @ E) & 1
    for i in range(r - 1, -1, -1):  # undo the row transpositions: A = P * (L E)
        LE[[i, P[i]]] = LE[[P[i], i]]
    assert np.array_equal(LE.astype(np.uint8), A.to_bits())


def test_solvers_at_scale_vs_reference_sha256():
    """mzd_ple (matrix + P + Q) and mzd_trsm_{lower,upper}_left at sizes where the reference recurses, against
    SHA-256 values of the real reference's results (tests/golden/solvers.json, make_golden.py --solvers)."""
    import hashlib
    import json
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solvers.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/solvers.json not generated")
    for e in json.load(open(path)):
        if e["what"] == "ple":
            m, n, seed = e["m"], e["n"], e["seed"]
            A = Mzd.random(m, n, seed)
            if e["kind"] == "lowrank":
                A = m4ri_amd.mzd_mul(None, Mzd.random(m, 5000, seed + 100), Mzd.random(5000, n, seed + 200), 0)
            elif e["kind"] == "zerocols":
                w = A.valid_words()
                w[:, :3] = 0
                w[:, 40:42] = 0
            r, P, Q = m4ri_amd.mzd_ple(A)
            h = hashlib.sha256(A.masked().tobytes() + P.astype(np.int32).tobytes() + Q.astype(np.int32).tobytes()).hexdigest()
            assert (r, h) == (e["rank"], e["sha256"]), e
        else:
            T, B = Mzd.random(e["m"], e["m"], e["seed"]), Mzd.random(e["m"], e["n"], e["seed"] + 1000)
            (m4ri_amd.mzd_trsm_upper_left if e["what"] == "trsm_upper" else m4ri_amd.mzd_trsm_lower_left)(T, B)
            assert hashlib.sha256(B.masked().tobytes()).hexdigest() == e["sha256"], e


def test_solver_edge_shapes(oracle):
    """Empty and one-line matrices through every solver entry point (the reference treats them as no-ops)."""
    for (m, n) in [(0, 0), (0, 5), (5, 0), (1, 1), (1, 200), (200, 1), (64, 64), (65, 1)]:
        A = Mzd.random(m, n, 3)
        Ao = A.copy()
        want = oracle.ple(Ao) if m and n else (0, np.arange(m, dtype=np.int32), np.arange(n, dtype=np.int32))
        got = m4ri_amd.mzd_ple(A)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and A.equal(Ao)
    for (mb, nb) in [(0, 0), (0, 7), (3, 0), (1, 70), (2, 1)]:
        T, B = Mzd.random(mb, mb, 4), Mzd.random(mb, nb, 5)
        for fn, ofn in ((m4ri_amd.mzd_trsm_lower_left, oracle.trsm_lower_left), (m4ri_amd.mzd_trsm_upper_left, oracle.trsm_upper_left)):
            X, Xo = B.copy(), B.copy()
            if mb and nb:
                ofn(T, Xo)
            fn(T, X)
            assert X.equal(Xo)
    # all-zero and identity inputs: rank 0 / full rank with no row operations
    Z = Mzd.init(300, 200)
    r, P, Q = m4ri_amd.mzd_ple(Z)
    assert r == 0 and np.array_equal(P, np.arange(300)) and np.array_equal(Q, np.arange(200)) and not Z.rows().any()
    I = Mzd.from_bits(np.eye(130, dtype=np.uint8))
    Io = I.copy()
    assert m4ri_amd.mzd_ple(I)[0] == 130 and I.equal(Io)
