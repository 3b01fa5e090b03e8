#!/usr/bin/env python3
"""usage: fresh_result_timing.py [n]  -- mzd_mul(NULL, A, B, 0) at n^3 (default 65536) against mzd_mul(C, A, B, 0) with a C the
caller already uses: what allocating the result costs inside the timed region of the reference's own bench
(bench/bench_multiplication.c:85-107).  Run once per strategy: M4RI_AMD_FRESH=populate|huge|lazy|zero with M4RI_AMD_RESULT_CACHE=0
(every result a NEW block), and with the cache on (the default)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m4ri_amd
from m4ri_amd.mzd import Mzd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m4ri_amd.init(0)
lib = m4ri_amd.lib()
A, B, C = Mzd.random(n, n, 3), Mzd.random(n, n, 4), Mzd.init(n, n)
C.buf.fill(0)
lib.mzd_mul(C.ptr, A.ptr, B.ptr, 0)
given, fresh = [], []
for _ in range(4):
    t = time.perf_counter()
    lib.mzd_mul(C.ptr, A.ptr, B.ptr, 0)
    given.append((time.perf_counter() - t) * 1e3)
want = C.valid_words().copy()
for k in range(5):
    t = time.perf_counter()
    r = lib.mzd_mul(None, A.ptr, B.ptr, 0)
    fresh.append((time.perf_counter() - t) * 1e3)
    R = m4ri_amd.from_struct_ptr(r, lib.m4ri_amd_result_free) if hasattr(m4ri_amd, "from_struct_ptr") else None
    if R is not None:
        assert (R.valid_words() == want).all(), "fresh result differs"
        del R
    else:
        lib.m4ri_amd_result_free(r)
print(f"n={n} FRESH={os.environ.get('M4RI_AMD_FRESH', 'populate')} CACHE={os.environ.get('M4RI_AMD_RESULT_CACHE', '1')}: "
      f"C given min {min(given):.1f} ms | C == NULL first {fresh[0]:.1f} ms, then {' '.join(f'{x:.1f}' for x in fresh[1:])} ms", flush=True)
