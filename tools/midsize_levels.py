#!/usr/bin/env python3
"""Would one more Strassen-Winograd level pay at sizes where the engine's default takes none / one?  Times resident products
under explicit cutoffs and prints the levels the engine chose."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import m4ri_amd

m4ri_amd.init(0)
for n in (4096, 6144, 8192, 10240, 12288, 16384, 20480, 24576):
    w = n // 64
    A = torch.zeros((n, w), dtype=torch.int64, device="cuda"); B = torch.zeros((n, w), dtype=torch.int64, device="cuda"); C = torch.zeros((n, w), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 3); m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 4)
    out = []
    for cutoff in (0, 1024, 2048, 3072, 4096, 6144):
        for _ in range(3):
            m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, cutoff=cutoff)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, cutoff=cutoff)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
        out.append(f"cutoff {cutoff}: L{m4ri_amd.get_stats().levels} {us:.0f} us")
    print(f"n={n}: " + " | ".join(out), flush=True)
