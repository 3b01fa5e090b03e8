#!/usr/bin/env python3
"""Device-resident product times over a fixed list of shapes (cubes from 512 to 32768, thin and flat ones, mul and addmul), each after
~50 ms of warm-up launches: the regression table for changes to the launch heuristics.  usage: many_shapes_timing.py [tag]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m4ri_amd.init(0)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = [(n, n, n) for n in (512, 1024, 2048, 3072, 4096, 6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576, 28672, 32768)] + [
    (512, 512, 65536), (1024, 1024, 65536), (65536, 1024, 1024), (1024, 65536, 1024), (4096, 65536, 4096), (8192, 32768, 8192), (16384, 4096, 16384),
    (24576, 8192, 8192), (32768, 4096, 32768), (8192, 8192, 131072), (131072, 8192, 8192), (16384, 16384, 65536), (65536, 16384, 16384), (1100, 1290, 1411),
    (6000, 6000, 6000), (9000, 9000, 9000), (20000, 20000, 20000), (30000, 30000, 30000)]
for (m, l, n) in shapes:
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
    C = torch.zeros((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
    m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
    out = []
    for add in (False, True):
        est = max(2e-5, m * l * n / 5e15)
        for _ in range(max(3, min(400, int(0.05 / est)))):
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
        torch.cuda.synchronize()
        reps = max(5, min(400, int(0.1 / est)))
        t = time.perf_counter()
        for _ in range(reps):
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / reps * 1e3)
    print(f"{tag} {m}x{l}x{n}: mul {out[0]:9.4f} ms  addmul {out[1]:9.4f} ms", flush=True)
    del A, B, C
