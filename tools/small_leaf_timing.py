#!/usr/bin/env python3
"""Device-resident SMALL products: the engine's plan (M4RI_AMD_SMALL_LEAF unset), the light one-launch kernel forced (=1) and never
(=0: pack + generation-4 leaf + reduce) -- run once per setting, the switch is read once per process.  Per shape: ms per product with the
launches back to back (throughput) and one at a time (latency), mul and addmul, and a checksum of C that must agree between settings.
usage: small_leaf_timing.py [tag]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m4ri_amd.init(0)
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("M4RI_AMD_SMALL_LEAF", "auto")
shapes = [(n, n, n) for n in (64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 2560, 3072, 4096, 5120)] + [
    (100, 1000, 100), (256, 4096, 256), (4096, 256, 4096), (4096, 4096, 256), (256, 256, 4096), (1100, 1290, 1411), (200, 8192, 8192), (8192, 200, 8192),
    (8192, 8192, 200), (64, 64, 65536), (65536, 64, 64), (2000, 3000, 100), (1024, 1024, 16384), (512, 512, 65536)]
for (m, l, n) in shapes:
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
    C = torch.zeros((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
    m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
    out, sums = [], []
    for add in (False, True):
        for _ in range(50):
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
        torch.cuda.synchronize()
        reps = 200
        t = time.perf_counter()
        for _ in range(reps):
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / reps * 1e6)
        t = time.perf_counter()
        for _ in range(reps):
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
            torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / reps * 1e6)
        C.zero_()
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
        torch.cuda.synchronize()
        w = torch.arange(1, C.numel() + 1, dtype=torch.int64, device="cuda").reshape(C.shape)
        sums.append(f"{int((C * w).sum().item()) & 0xffffffffffff:012x}")
    st = m4ri_amd.get_stats()
    print(f"{tag} {m}x{l}x{n}: mul {out[0]:7.1f} us back to back, {out[1]:7.1f} one at a time | addmul {out[2]:7.1f} / {out[3]:7.1f} | gen {st.leaf_gen} levels {st.levels} | {sums[0]} {sums[1]}", flush=True)
    del A, B, C
