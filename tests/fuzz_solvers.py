#!/usr/bin/env python3
"""Fuzz of the decomposition / echelon routines against the oracle on random shapes and structures (a script, not a
pytest module; it lives under tests/ because it runs the checker; GPU box): PLE and PLUQ in both flavours (identity or the reference's recursion leftovers behind the rank), both echelon
forms, the column permutations, triangular solves (also above 4096 rows), triangular inverses, transposes, inverses.  usage: fuzz_solvers.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402
import cpu_libs  # noqa: E402


def structured(rng, m, n):
    A = Mzd.random(m, n, int(rng.integers(1, 1 << 30)))
    w = A.valid_words()
    kind = rng.integers(0, 6)
    if kind == 1:    # zero word columns
        for _ in range(rng.integers(1, 4)):
            w[:, rng.integers(0, w.shape[1])] = 0
    elif kind == 2:  # repeated rows / zero rows
        k = int(rng.integers(1, max(2, m // 3)))
        w[m - k:] = w[:k]
        w[m - int(rng.integers(0, k + 1)):] = 0
    elif kind == 3:  # sparse
        for _ in range(3):
            w &= Mzd.random(m, n, int(rng.integers(1, 1 << 30))).valid_words()
    elif kind == 4 and m > 2 and n > 2:  # low rank
        r = int(rng.integers(1, max(2, min(m, n) // 2)))
        A = m4ri_amd.mzd_mul(None, Mzd.random(m, r, int(rng.integers(1, 1 << 30))), Mzd.random(r, n, int(rng.integers(1, 1 << 30))), 0)
    elif kind == 5:  # columns copying their left neighbour
        for c in rng.choice(np.arange(1, n), size=max(1, n // 30), replace=False) if n > 1 else []:
            c = int(c)
            bit = (w[:, (c - 1) // 64] >> np.uint64((c - 1) % 64)) & np.uint64(1)
            w[:, c // 64] = (w[:, c // 64] & ~(np.uint64(1) << np.uint64(c % 64))) | (bit << np.uint64(c % 64))
    return A, int(kind)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc = cpu_libs.oracle()
    m4ri_amd.init(0)
    t0, cases, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        big = rng.random() < 0.15
        m = int(rng.integers(1, 2600 if big else 400))
        n = int(rng.integers(1, 2600 if big else 400))
        if big and rng.random() < 0.5:  # above the reference's recursion cutoff (width * rows > 524288 words)
            m, n = int(rng.integers(3000, 5000)), int(rng.integers(9000, 12000))
        A, kind = structured(rng, m, n)
        what = ["ple", "ple_rec", "pluq", "pluq_rec", "ech0", "ech1", "perm", "trsm", "trsm_big", "trtri", "transpose", "inv"][int(rng.integers(0, 12))]
        ok = True
        if what in ("ple", "ple_rec", "pluq", "pluq_rec"):
            pluq, rec = what.startswith("pluq"), what.endswith("rec")
            name = ("_mzd_pluq" if rec else "_mzd_pluq_russian") if pluq else ("_mzd_ple" if rec else "_mzd_ple_russian")
            Ao, Ag = A.copy(), A.copy()
            want, got = orc.ple(Ao, pluq=pluq, recursive=rec), m4ri_amd.mzd_ple(Ag, 0, name)
            ok = got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and np.array_equal(Ag.valid_words(), Ao.valid_words())
        elif what in ("ech0", "ech1"):
            full = int(what[-1])
            Ao, Ag = A.copy(), A.copy()
            which = ["mzd_echelonize", "mzd_echelonize_m4ri", "mzd_echelonize_pluq"][int(rng.integers(0, 3))]
            ok = m4ri_amd.mzd_echelonize(Ag, full, which) == orc.echelonize(Ao, full) and np.array_equal(Ag.valid_words(), Ao.valid_words())
        elif what == "perm":
            Q = np.array([rng.integers(i, n) for i in range(n)], dtype=np.int32)
            for trans in (False, True):
                Ao, Ag = A.copy(), A.copy()
                orc.apply_p_right(Ao, Q, trans)
                m4ri_amd.mzd_apply_p_right(Ag, Q, trans)
                ok = ok and np.array_equal(Ag.valid_words(), Ao.valid_words())
        elif what == "trsm_big":  # more than 4096 rows: the 4096-row block inverses, ragged last block
            mb, nb = int(rng.integers(4097, 9500)), int(rng.integers(1, 260))
            T = Mzd.random(mb, mb, int(rng.integers(1, 1 << 30)))
            B = Mzd.random(mb, nb, int(rng.integers(1, 1 << 30)))
            upper = bool(rng.integers(0, 2))
            Bo, Bg = B.copy(), B.copy()
            (orc.trsm_upper_left if upper else orc.trsm_lower_left)(T, Bo)
            (m4ri_amd.mzd_trsm_upper_left if upper else m4ri_amd.mzd_trsm_lower_left)(T, Bg)
            ok = np.array_equal(Bg.valid_words(), Bo.valid_words())
        elif what == "trtri":  # unit diagonal, the lower triangle junk that must survive
            k = min(max(m, n), 2200)
            U = Mzd.random(k, k, int(rng.integers(1, 1 << 30)))
            idx = np.arange(k)
            U.valid_words()[idx, idx // 64] |= np.uint64(1) << (idx % 64).astype(np.uint64)
            Uo, Ug = U.copy(), U.copy()
            orc.trtri_upper(Uo)
            m4ri_amd.mzd_trtri_upper(Ug, ["mzd_trtri_upper", "mzd_trtri_upper_russian"][int(rng.integers(0, 2))])
            ok = np.array_equal(Ug.valid_words(), Uo.valid_words())
        elif what == "transpose":
            ok = np.array_equal(m4ri_amd.mzd_transpose(A).valid_words(), orc.transpose(A).valid_words())
        elif what == "inv":  # singular inputs (most structured ones) take the reference's augmented elimination, invertible ones the PLUQ road
            k = min(m, 900)
            if rng.random() < 0.5:
                Lm, Um = Mzd.random(k, k, int(rng.integers(1, 1 << 30))).to_bits(), Mzd.random(k, k, int(rng.integers(1, 1 << 30))).to_bits()
                Lm, Um = np.tril(Lm), np.triu(Um)
                np.fill_diagonal(Lm, 1); np.fill_diagonal(Um, 1)
                S = m4ri_amd.mzd_mul(None, Mzd.from_bits(Lm), Mzd.from_bits(Um), 0)
            else:
                S = structured(rng, k, k)[0]
            ok = np.array_equal(m4ri_amd.mzd_inv_m4ri(S).valid_words(), orc.inv(S).valid_words())
        else:
            mb, nb = min(m, 700), min(n, 900)
            T = Mzd.random(mb, mb, int(rng.integers(1, 1 << 30)))
            B = Mzd.random(mb, nb, int(rng.integers(1, 1 << 30)))
            for upper in (False, True):
                Bo, Bg = B.copy(), B.copy()
                (orc.trsm_upper_left if upper else orc.trsm_lower_left)(T, Bo)
                (m4ri_amd.mzd_trsm_upper_left if upper else m4ri_amd.mzd_trsm_lower_left)(T, Bg)
                ok = ok and np.array_equal(Bg.valid_words(), Bo.valid_words())
        cases += 1
        if not ok:
            bad += 1
            print(f"MISMATCH {what} m={m} n={n} kind={kind}", flush=True)
    print(f"fuzz: {cases} cases in {time.time() - t0:.0f} s, {bad} mismatches")
    print("FUZZ OK" if not bad else "FUZZ FAILED")
    return bad


if __name__ == "__main__":
    sys.exit(main())
