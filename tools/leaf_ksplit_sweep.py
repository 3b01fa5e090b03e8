#!/usr/bin/env python3
"""One leaf launch (mzd_mul_m4rm on resident operands) at forced inner-dimension splits beside the split the engine picks:
usage: leaf_ksplit_sweep.py m l n [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m, l, n = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
m4ri_amd.init(0)
wl, wn = (l + 63) // 64, (n + 63) // 64
A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
C = torch.empty((m, wn), dtype=torch.int64, device="cuda")
m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
out, sums = [], []
for ks in (0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 0):
    for _ in range(20):   # (enough launches for the clocks to settle: the first configuration of a run measured 5 - 10 % slow otherwise)
        m4ri_amd.m4rm_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, ksplit=ks)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        m4ri_amd.m4rm_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, ksplit=ks)
    torch.cuda.synchronize()
    out.append(f"ksplit {ks if ks else 'auto'}: {(time.perf_counter() - t) / reps * 1e3:.3f} ms")
    sums.append(int(C.sum().item()))
print(f"{m}x{l}x{n}: " + " | ".join(out) + (" | results agree" if len(set(sums)) == 1 else " | RESULTS DIFFER"))
