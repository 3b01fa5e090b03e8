#!/usr/bin/env python3
"""Timeline of the block pipeline behind mzd_mul on host matrices (mzd_api.hip run_pipelined, M4RI_AMD_PIPE_TRACE=1): when every
block goes up, every product runs and every block of C comes down.  usage: host_pipeline_trace.py [n [calls]]   (run on the GPU box)"""
import os
import sys
import time

os.environ["M4RI_AMD_PIPE_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m4ri_amd
from m4ri_amd.mzd import Mzd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m4ri_amd.init(0)
A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
C = Mzd.init(n, n)
for i in range(calls):
    t = time.perf_counter()
    m4ri_amd.mzd_mul(C, A, B, 0)
    print(f"call {i}: {1e3 * (time.perf_counter() - t):.2f} ms", file=sys.stderr, flush=True)
