#!/usr/bin/env python3
"""usage: time_product.py m l n [reps] [warmup] -- one resident product C = A*B (m4ri_amd_mul_dev) timed over `reps` back-to-back
calls after `warmup`; prints ms per product, the leaf launch's share (HIP events) and a checksum of C, so that runs of one binary
under different developer switches (environment) can be compared line by line."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m, l, n = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
warm = int(sys.argv[5]) if len(sys.argv) > 5 else 10
tag = os.environ.get("TAG", "")
m4ri_amd.init(0)
wl, wn = (l + 63) // 64, (n + 63) // 64
A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
C = torch.empty((m, wn), dtype=torch.int64, device="cuda")
m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
call = lambda: m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n)
for _ in range(warm):
    call()
torch.cuda.synchronize()
m4ri_amd.set_profiling(2)
t = time.perf_counter()
for _ in range(reps):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
st = m4ri_amd.get_stats()
leaf = st.cum_leaf_ms / max(1, reps)
print(f"{tag:28s} {m}x{l}x{n}: {dt * 1e3:8.3f} ms/product, leaf launches {leaf:8.3f} ms, levels {st.levels}, checksum {int(C.sum().item()) & 0xffffffffffff:012x}", flush=True)
