"""Developer probe: N calls of mzd_mul on small host matrices (run under rocprofv3 --kernel-trace --stats to see what a call launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import m4ri_amd
from m4ri_amd.mzd import Mzd
n = int(sys.argv[1]); reps = int(sys.argv[2])
m4ri_amd.init(0)
m4ri_amd.set_small_product_threshold(0)
A, B, C = Mzd.random(n, n, 3), Mzd.random(n, n, 4), Mzd.init(n, n)
m4ri_amd.mzd_mul(C, A, B, 0)
t = time.perf_counter()
for _ in range(reps):
    m4ri_amd.mzd_mul(C, A, B, 0)
print(f"n={n}: {(time.perf_counter() - t) / reps * 1e6:.1f} us per call")
