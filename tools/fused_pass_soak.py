#!/usr/bin/env python3
"""Soak of the fused Winograd passes: random shapes / depths / strides / accumulate, the product with 4-, 3- and 2-level passes against
the same product with single-level passes (itself checked against the oracle by tests/test_gpu_parity.py).  Device only, so the shapes
can be large enough to reach every form of the passes (LDS forms want leaf rows of 16 / 32 words).  usage: fused_pass_soak.py [cases [seed]]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import m4ri_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
m4ri_amd.init(0)
bad = 0
for case in range(cases):
    L = int(rng.integers(1, 5))
    g = 1 << L
    # leaf shapes: rows 32..1024 (x 1..3 odd multiples), leaf row words from {1,2,5,8,16,20,32,48,64}
    mrows = int(rng.choice([32, 64, 96, 160, 256, 512, 1024])) * g
    lw = int(rng.choice([1, 2, 5, 8, 16, 20, 32, 48]))
    nw = int(rng.choice([1, 2, 5, 8, 16, 32, 64]))
    m, l, n = mrows + int(rng.integers(0, 2)) * int(rng.integers(1, 40)), lw * 64 * g + int(rng.integers(0, 2)) * int(rng.integers(1, 200)), nw * 64 * g + int(rng.integers(0, 2)) * int(rng.integers(1, 200))
    if m * l * n > 2 ** 41:
        continue
    add = bool(rng.integers(0, 2))
    pad_a, pad_b, pad_c = (int(x) for x in rng.integers(0, 3, 3) * rng.integers(1, 7, 3))
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.zeros((m, wl + pad_a), dtype=torch.int64, device="cuda")
    B = torch.zeros((l, wn + pad_b), dtype=torch.int64, device="cuda")
    C0 = torch.zeros((m, wn + pad_c), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl + pad_a, m, l, 1000 + case)
    m4ri_amd.fill_dev(B.data_ptr(), wn + pad_b, l, n, 2000 + case)
    m4ri_amd.fill_dev(C0.data_ptr(), wn + pad_c, m, n, 3000 + case)
    if pad_c:
        C0[:, wn:] = -1
    os.environ["M4RI_AMD_LEVELS"] = str(L)
    outs = {}
    for fuse in (1, 2, 3, 4):
        m4ri_amd.set_max_fuse(fuse)
        C = C0.clone()
        m4ri_amd.mul_dev(C.data_ptr(), wn + pad_c, A.data_ptr(), wl + pad_a, B.data_ptr(), wn + pad_b, m, l, n, add, 0)
        torch.cuda.synchronize()
        outs[fuse] = C
    lv = m4ri_amd.get_stats().levels
    ok = all(torch.equal(outs[1], outs[f]) for f in (2, 3, 4)) and (not pad_c or bool((outs[4][:, wn:] == -1).all()))
    if not ok:
        bad += 1
    print(f"case {case}: {m}x{l}x{n} L={lv} add={add} pads {pad_a},{pad_b},{pad_c}: {'ok' if ok else 'MISMATCH'}", flush=True)
os.environ.pop("M4RI_AMD_LEVELS", None)
m4ri_amd.set_max_fuse(4)
print(f"FUSED_PASS_SOAK {'ok' if bad == 0 else 'FAILED'}: {bad} mismatches")
sys.exit(1 if bad else 0)
