#!/usr/bin/env python3
"""Race hunt for the solver kernels: the same PLE / TRSM repeated, every result hashed -- all runs of a case must agree
(the pivot search is a single workgroup with LDS atomics and a sliding window; the elimination passes overlap on one
stream).  usage: stress_solvers.py [repeats]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    m4ri_amd.init(0)
    bad = 0
    for (m, n, seed, sparse) in [(5000, 5000, 1, False), (70000, 448, 2, True), (3000, 9000, 3, False), (16384, 16384, 4, False), (9000, 2000, 5, True)]:
        A0 = Mzd.random(m, n, seed)
        if sparse:  # pivots far down: the slow path of the pivot search (rows beyond the window, stale lanes catching up)
            for k in range(4):
                A0.valid_words()[:, :] &= Mzd.random(m, n, seed + 10 + k).valid_words()
        seen = set()
        for _ in range(reps):
            A = A0.copy()
            r, P, Q = m4ri_amd.mzd_ple(A)
            seen.add(hashlib.sha256(A.masked().tobytes() + P.tobytes() + Q.tobytes()).hexdigest())
        print(f"ple {m}x{n} sparse={sparse}: {reps} runs, {len(seen)} distinct result(s), rank {r}", flush=True)
        bad += len(seen) != 1
    for (mb, nb, seed) in [(4096, 8192, 7), (10000, 3000, 8)]:
        T, B0 = Mzd.random(mb, mb, seed), Mzd.random(mb, nb, seed + 1)
        for upper in (False, True):
            seen = set()
            for _ in range(reps):
                B = B0.copy()
                (m4ri_amd.mzd_trsm_upper_left if upper else m4ri_amd.mzd_trsm_lower_left)(T, B)
                seen.add(hashlib.sha256(B.masked().tobytes()).hexdigest())
            print(f"trsm {'upper' if upper else 'lower'} {mb}x{nb}: {reps} runs, {len(seen)} distinct result(s)", flush=True)
            bad += len(seen) != 1
    # the panel step of the PLE switched on by size, the 4096-row block inverses of the TRSM, transposes, triangular inverses
    big = max(3, reps // 8)
    A0 = Mzd.random(45056, 45056, 21)
    seen = set()
    for _ in range(big):
        A = A0.copy()
        r, P, Q = m4ri_amd.mzd_ple(A, 0, "mzd_pluq")
        seen.add(hashlib.sha256(A.masked().tobytes() + P.tobytes() + Q.tobytes()).hexdigest())
    print(f"pluq 45056^2 (panels on): {big} runs, {len(seen)} distinct result(s), rank {r}", flush=True)
    bad += len(seen) != 1
    T, B0 = Mzd.random(20000, 20000, 22), Mzd.random(20000, 9000, 23)
    for name, fn in (("trsm upper 20000x9000", lambda X: m4ri_amd.mzd_trsm_upper_left(T, X)), ("transpose 20000x9000", lambda X: m4ri_amd.mzd_transpose(X))):
        seen = set()
        for _ in range(big * 2):
            out = fn(B0.copy())
            seen.add(hashlib.sha256(out.masked().tobytes()).hexdigest())
        print(f"{name}: {big * 2} runs, {len(seen)} distinct result(s)", flush=True)
        bad += len(seen) != 1
    seen = set()
    for _ in range(big * 2):
        U = T.copy()
        m4ri_amd.mzd_trtri_upper(U)
        seen.add(hashlib.sha256(U.masked().tobytes()).hexdigest())
    print(f"trtri_upper 20000: {big * 2} runs, {len(seen)} distinct result(s)", flush=True)
    bad += len(seen) != 1
    print("STRESS OK" if not bad else "STRESS FAILED")
    return bad


if __name__ == "__main__":
    sys.exit(main())
