"""The oracle against the committed golden fixtures (generated from the real reference by
tests/golden/make_golden.py).  CPU only; this is what pins the oracle on the GPU box, where
/root/reference does not exist."""
import os

import numpy as np
import pytest

from m4ri_amd.mzd import Mzd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kats():
    z = np.load(os.path.join(GOLD, "kat_small.npz"))
    for i in range(int(z["ncases"])):
        m, l, n, par = (int(x) for x in z[f"case{i}_meta"])
        yield i, str(z[f"case{i}_op"]), m, l, n, par, z[f"case{i}_seeds"], z[f"case{i}_A"], z[f"case{i}_B"], z[f"case{i}_C"]


def mat_from_words(rows, cols, words):
    M = Mzd.init(rows, cols)
    if rows and cols:
        M.valid_words()[:, :] = words
    return M


def run_kat(mul, addmul, m4rm, op, m, l, n, par, seeds, Aw, Bw):
    A = mat_from_words(m, l, Aw)
    # the stored inputs ARE the seeded fill (fixture self-consistency)
    assert np.array_equal(A.masked(), Mzd.random(m, l, int(seeds[0])).masked())
    B = A if op == "sqr" else mat_from_words(l, n, Bw)
    if op in ("mul", "sqr"):
        return mul(None, A, B, par)
    if op == "m4rm":
        return m4rm(Mzd.init(m, n), A, B, par)
    return addmul(Mzd.random(m, n, int(seeds[2])), A, B, par)


@pytest.mark.parametrize("case", list(load_kats()), ids=lambda c: f"{c[1]}-{c[2]}x{c[3]}x{c[4]}-{c[5]}")
def test_oracle_reproduces_kat(oracle, case):
    i, op, m, l, n, par, seeds, Aw, Bw, Cw = case
    got = run_kat(oracle.mul, oracle.addmul, lambda C, A, B, k: oracle.mul_m4rm(C, A, B, k, 1), op, m, l, n, par, seeds, Aw, Bw)
    assert np.array_equal(got.masked(), Cw)


def test_oracle_reproduces_fingerprints(oracle):
    z = np.load(os.path.join(GOLD, "fingerprints.npz"))
    for op, (m, l, n, par), (sa, sb, sc), fp in zip(z["ops"], z["meta"], z["seeds"], z["fp"]):
        m, l, n, par = int(m), int(l), int(n), int(par)
        if m * l * n > 4096 ** 3:
            continue  # the big ones are GPU-side fixtures; the oracle would need minutes
        A, B = Mzd.random(m, l, int(sa)), Mzd.random(l, n, int(sb))
        if op == "mul":
            C = oracle.mul(None, A, B, par)
        elif op == "m4rm":
            C = oracle.mul_m4rm(Mzd.init(m, n), A, B, par, 1)
        else:
            C = oracle.addmul(Mzd.random(m, n, int(sc)), A, B, par)
        assert oracle.fingerprint(C) == int(fp), (op, m, l, n, par)
