"""One resident mzd_trsm_upper_left at n (default 65536), for rocprofv3 --kernel-trace (tools/README.md)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m4ri_amd
from m4ri_amd.mzd import Mzd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m4ri_amd.init(0)
T, B = Mzd.random(n, n, 3), Mzd.random(n, n, 2)
m4ri_amd.pin(T); m4ri_amd.pin(B)
m4ri_amd.mzd_trsm_upper_left(T, B)
t = time.perf_counter()
m4ri_amd.mzd_trsm_upper_left(T, B)
print(f"trsm_upper_left n={n}: {(time.perf_counter() - t) * 1e3:.1f} ms")
