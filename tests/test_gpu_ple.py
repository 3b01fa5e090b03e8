"""PLE decomposition on the GPU (include/m4ri_amd.h: mzd_ple / _mzd_ple / _mzd_ple_russian; reference
m4ri/ple.c:33-171, m4ri/ple_russian.c:380-617) against the oracle's column-by-column PLE, which
tests/test_ple_oracle.py pins to the reference: decomposed matrix, P, Q and rank, bit for bit."""
import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd
from test_ple_oracle import RECURSIVE_CASES, SHAPES, _defects, _make

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
    want = oracle.ple(Ao, recursive=True)
    _same(m4ri_amd.mzd_ple(Ag), want, Ag, Ao)


@pytest.mark.parametrize("m,n", SHAPES)
@pytest.mark.parametrize("kind", ["random", "lowrank", "sparse", "zerocols"])
def test_pluq_matches_oracle(oracle, m, n, kind):
    """mzd_pluq / _mzd_pluq / _mzd_pluq_russian (ple.c:41-60, ple_russian.c:625-629): PLE + the column step."""
    A = _make(kind, m, n, 2000 + 7 * m + n)
    Ao = A.copy()
    want = oracle.ple(Ao, pluq=True)
    for which in ("mzd_pluq", "_mzd_pluq", "_mzd_pluq_russian"):
        Ag = A.copy()
        _same(m4ri_amd.mzd_ple(Ag, 0, which), want, Ag, Ao)


@pytest.mark.parametrize("m,n,kind", [(5000, 5000, "random"), (3000, 9000, "random"), (6000, 6000, "lowrank"), (4099, 8200, "zerocols"),
                                     (9000, 3000, "lowrank"), (40, 530000, "random")])
def test_larger_pluq_matches_oracle(oracle, m, n, kind):
    """Several row groups of the column step; the last shape is wider than the LDS row copy (global row copies)."""
    if kind == "lowrank":
        A = m4ri_amd.mzd_mul(None, Mzd.random(m, 1500, 5), Mzd.random(1500, n, 6), 0)
    else:
        A = _make(kind, m, n, 78)
    if n > 100000:  # push pivots far to the right: only a few columns carry data
        w = A.valid_words()
        w[:, : w.shape[1] - 3] = 0
        w[:, 5] = Mzd.random(m, 64, 3).valid_words()[:, 0]
    Ao, Ag = A.copy(), A.copy()
    want = oracle.ple(Ao, pluq=True, recursive=True)
    _same(m4ri_amd.mzd_ple(Ag, 0, "mzd_pluq"), want, Ag, Ao)


@pytest.mark.parametrize("m,n", [(1, 2), (5, 64), (64, 65), (100, 300), (300, 100), (700, 1000), (2000, 1500)])
def test_apply_p_right_trans_tri(m, n):
    """mzd_apply_p_right_trans_tri (mzp.c:279-293) with arbitrary transpositions Q[i] >= i: row r takes the swaps
    i > r in ascending order -- replayed with numpy on the bit matrix."""
    rng = np.random.default_rng(m * 1000 + n)
    A = Mzd.random(m, n, 11)
    Q = np.array([rng.integers(i, n) if rng.random() < 0.7 else i for i in range(n)], dtype=np.int32)
    b = A.to_bits()
    for i in range(n):
        if Q[i] != i:
            rows = slice(0, min(m, i))
            b[rows, [i, Q[i]]] = b[rows, [Q[i], i]]
    m4ri_amd.mzd_apply_p_right_trans_tri(A, Q)
    assert np.array_equal(A.to_bits(), b)


@pytest.mark.parametrize("m,n,dup,zero", RECURSIVE_CASES)
def test_recursive_flavours_match_oracle(oracle, m, n, dup, zero):
    """Above __M4RI_PLE_CUTOFF mzd_ple / _mzd_ple / mzd_pluq / _mzd_pluq leave the transpositions of the reference's
    column-halving recursion in Q behind the rank (and mzd_pluq applies them); _mzd_ple_russian / _mzd_pluq_russian leave
    the identity.  Both against the oracle's two restatements, which tests/test_ple_oracle.py pins to the reference."""
    A = _defects(m, n, 3000 + m + n, dup, zero)
    for pluq in (False, True):
        Ao, Af = A.copy(), A.copy()
        want_rec, want_flat = oracle.ple(Ao, pluq=pluq, recursive=True), oracle.ple(Af, pluq=pluq)
        for which in (("mzd_pluq", "_mzd_pluq") if pluq else ("mzd_ple", "_mzd_ple")):
            Ag = A.copy()
            _same(m4ri_amd.mzd_ple(Ag, 0, which), want_rec, Ag, Ao)
        Ag = A.copy()
        _same(m4ri_amd.mzd_ple(Ag, 0, "_mzd_pluq_russian" if pluq else "_mzd_ple_russian"), want_flat, Ag, Af)


def test_pluq_on_a_window_keeps_the_parent(oracle):
    P0 = Mzd.random(900, 1000, 9)
    P0.valid_words()[:, 2] = 0
    for (r0, c0, m, n) in [(10, 64, 500, 333), (0, 0, 900, 130), (100, 128, 64, 64), (3, 0, 300, 1000)]:
        Po, Pg = Mzd(900, 1000, buf=P0.buf.copy()), Mzd(900, 1000, buf=P0.buf.copy())
        wo, wg = Po.window(r0, c0, r0 + m, c0 + n), Pg.window(r0, c0, r0 + m, c0 + n)
        want = oracle.ple(wo, pluq=True)
        got = m4ri_amd.mzd_ple(wg, 0, "mzd_pluq")
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        assert np.array_equal(Po.buf, Pg.buf)


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
    assert os.path.exists(path), "committed fixture tests/golden/solvers.json is missing"
    for e in json.load(open(path)):
        if e["what"] in ("ple", "pluq"):
            m, n, seed = e["m"], e["n"], e["seed"]
            A = Mzd.random(m, n, seed)
            if e["kind"] == "lowrank":
                A = m4ri_amd.mzd_mul(None, Mzd.random(m, 5000, seed + 100), Mzd.random(5000, n, seed + 200), 0)
            elif e["kind"] == "zerocols":
                w = A.valid_words()
                w[:, :3] = 0
                w[:, 40:42] = 0
            elif e["kind"] == "defects":
                A = _defects(m, n, seed, m // 16, m // 64)
            r, P, Q = m4ri_amd.mzd_ple(A, 0, "mzd_" + e["what"])
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


@pytest.mark.parametrize("panel_words", [1, 2, 5])
def test_panel_step_forced_on(panel_words):
    """The panel step of the PLE (ple.hip: blocks update only their own panel, the rest of the matrix once per panel through a
    TRSM and an engine product) switches itself on from about 150 MiB of matrix; here the whole parity suite of this file
    runs again in a child process with panels of 64 / 128 / 320 columns forced on (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, M4RI_AMD_PLE_PANELS="1", M4RI_AMD_PLE_PANEL=str(panel_words))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_ple.py"), "-x", "-q", "-m", "gpu", "-k", "not forced_on and not at_scale",
                        "-p", "no:cacheprovider"], cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout
