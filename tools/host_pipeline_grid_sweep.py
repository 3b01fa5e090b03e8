#!/usr/bin/env python3
"""mzd_mul from host memory at one size for the block grid given in M4RI_AMD_PIPE_GRID (run once per grid: the library
reads the variable once).  usage: M4RI_AMD_PIPE_GRID=4,2 host_pipeline_grid_sweep.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m4ri_amd
from m4ri_amd.mzd import Mzd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m4ri_amd.init(0)
if len(sys.argv) > 2:  # second argument: the pipeline's size threshold in MiB (0 = off)
    m4ri_amd.set_host_pipeline(int(sys.argv[2]) << 20)
A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
C = Mzd.init(n, n)
m4ri_amd.mzd_mul(C, A, B, 0)
ts = []
for _ in range(5):
    t = time.perf_counter()
    m4ri_amd.mzd_mul(C, A, B, 0)
    ts.append(time.perf_counter() - t)
print(f"grid {os.environ.get('M4RI_AMD_PIPE_GRID', 'default')}: n={n} best {min(ts) * 1e3:.2f} ms median {sorted(ts)[2] * 1e3:.2f} ms", flush=True)
