"""mzd_echelonize on sparse random matrices (the reference's bench/bench_elimination_sparse.c draws every bit with a given
density): the pivot search leaves its one-wave fast path when pivots lie more than 128 rows down.  usage: [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import m4ri_amd
from m4ri_amd.mzd import Mzd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m4ri_amd.init(0)
rng = np.random.default_rng(5)
for density in (0.5, 0.1, 0.01, 0.001, 0.0002):
    A = Mzd(n, n)
    w = A.valid_words()
    if density == 0.5:
        w[:] = Mzd.random(n, n, 1).valid_words()
    else:
        k = int(n * n * density)
        r, c = rng.integers(0, n, k), rng.integers(0, n, k)
        np.bitwise_or.at(w, (r, c // 64), np.uint64(1) << (c % 64).astype(np.uint64))
    for full in (0, 1):
        B = A.copy()
        m4ri_amd.pin(B)
        t = time.perf_counter()
        rank = m4ri_amd.mzd_echelonize(B, full)
        dt = time.perf_counter() - t
        m4ri_amd.unpin(B)
        print(f"n={n} density {density:g} full={full}: {dt * 1e3:9.1f} ms  rank {rank}", flush=True)
