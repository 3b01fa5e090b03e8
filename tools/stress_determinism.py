#!/usr/bin/env python3
"""Race hunt: the same products over and over, every result compared bit for bit with the first one
(a missed barrier or an LDS double-buffering race shows up as a run that differs).  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m4ri_amd.init(0)
bad = 0
CASES = [(65536, 65536, 65536, 0, 25), (32768, 65536, 32768, 0, 40), (16384, 65536, 32768, 0, 60), (8192, 8192, 8192, 0, 300),
         (4096, 4096, 4096, 0, 300), (12288, 4096, 51200, 0, 100), (5000, 3000, 7000, 1024, 200), (2048, 2048, 2048, 0, 300),
         (1000, 1000, 1000, 256, 300), (131072, 8192, 131072, 0, 8)]
for m, l, n, cutoff, reps in CASES:
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 11)
    m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 12)
    ref = None
    diffs = 0
    for r in range(reps):
        C = torch.full((m, wn), -1, dtype=torch.int64, device="cuda")  # dirty result buffer
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, cutoff=cutoff)
        torch.cuda.synchronize()
        if ref is None:
            ref = C
        elif not torch.equal(ref, C):
            diffs += 1
    print(f"{m} x {l} x {n} cutoff {cutoff}: {reps} runs, {diffs} differ from the first", flush=True)
    bad += diffs
    del A, B, C, ref
    m4ri_amd.lib().m4ri_amd_release_workspace()
print("STRESS", "FAILED" if bad else "ALL IDENTICAL")
sys.exit(1 if bad else 0)
