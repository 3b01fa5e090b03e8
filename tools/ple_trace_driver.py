"""Developer probe: one resident mzd_ple of n x n (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import m4ri_amd
from m4ri_amd.mzd import Mzd
n = int(sys.argv[1])
m4ri_amd.init(0)
for rep in range(2):
    A = Mzd.random(n, n, 1)
    m4ri_amd.pin(A)
    t = time.perf_counter()
    r = m4ri_amd.mzd_ple(A)
    dt = time.perf_counter() - t
    m4ri_amd.unpin(A)
print(f"n={n}: {dt * 1e3:.2f} ms rank {r[0]}")
