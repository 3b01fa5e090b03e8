#!/usr/bin/env python3
"""Single short products (the TRSM's block products, small user products) under each leaf generation:
M4RI_AMD_LEAF_GEN=4|3|1 forces the kernel; run once per setting."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import m4ri_amd

m4ri_amd.init(0)
SHAPES = [(64, 64, 65536), (128, 128, 65536), (128, 128, 1024), (160, 160, 65536), (192, 192, 65536), (192, 192, 1024), (256, 256, 65536), (256, 256, 1024), (100, 4096, 4096), (200, 4096, 4096), (300, 4096, 4096), (512, 512, 65536), (1024, 1024, 65536), (1100, 1100, 1100), (4400, 4400, 4400), (6000, 6000, 6000), (8800, 8800, 8800), (16384, 16384, 16384)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (m, l, n) in SHAPES:
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.zeros((m, wl), dtype=torch.int64, device="cuda"); B = torch.zeros((l, wn), dtype=torch.int64, device="cuda"); C = torch.zeros((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3); m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
    for _ in range(5):
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    fp = int(C.view(-1)[::97].sum().item()) & 0xffffffff
    print(f"gen {os.environ.get('M4RI_AMD_LEAF_GEN', 'auto')}: {m} x {l} x {n}: {us:8.1f} us  fp={fp:08x}", flush=True)
