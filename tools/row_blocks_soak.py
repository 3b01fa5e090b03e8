#!/usr/bin/env python3
"""Soak of the engine's own plans on big ragged shapes: the product as planned (rows in blocks, each at its depth) against ONE product of
the same operands at a forced depth (M4RI_AMD_LEVELS: another schedule, other kernels for the strips), with and without accumulate and
strided parents.  Device only.  usage: row_blocks_soak.py [cases [seed]]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import m4ri_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
m4ri_amd.init(0)
bad = split = 0
for case in range(cases):
    m = int(rng.choice([rng.integers(8192, 80000), 4096 * rng.integers(3, 18) + rng.integers(0, 3) * rng.integers(1, 600)]))
    l = int(rng.choice([rng.integers(4096, 40000), 1024 * rng.integers(4, 36) + rng.integers(0, 2) * rng.integers(1, 900)]))
    n = int(rng.choice([rng.integers(4096, 40000), 1024 * rng.integers(4, 36) + rng.integers(0, 2) * rng.integers(1, 900)]))
    add = bool(rng.integers(0, 2))
    pad_a, pad_b, pad_c = (int(x) for x in rng.integers(0, 2, 3) * rng.integers(1, 9, 3))
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.zeros((m, wl + pad_a), dtype=torch.int64, device="cuda")
    B = torch.zeros((l, wn + pad_b), dtype=torch.int64, device="cuda")
    C0 = torch.zeros((m, wn + pad_c), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl + pad_a, m, l, 5000 + case)
    m4ri_amd.fill_dev(B.data_ptr(), wn + pad_b, l, n, 6000 + case)
    m4ri_amd.fill_dev(C0.data_ptr(), wn + pad_c, m, n, 7000 + case)
    if pad_c:
        C0[:, wn:] = -1
    os.environ.pop("M4RI_AMD_LEVELS", None)
    plan = m4ri_amd.plan_row_blocks(m, l, n)
    split += len(plan) > 1
    C = C0.clone()
    m4ri_amd.mul_dev(C.data_ptr(), wn + pad_c, A.data_ptr(), wl + pad_a, B.data_ptr(), wn + pad_b, m, l, n, add, 0)
    torch.cuda.synchronize()
    other = int(rng.integers(0, 4))
    os.environ["M4RI_AMD_LEVELS"] = str(other)
    D = C0.clone()
    m4ri_amd.mul_dev(D.data_ptr(), wn + pad_c, A.data_ptr(), wl + pad_a, B.data_ptr(), wn + pad_b, m, l, n, add, 0)
    torch.cuda.synchronize()
    lv = m4ri_amd.get_stats().levels
    ok = torch.equal(C, D) and (not pad_c or bool((C[:, wn:] == -1).all()))
    bad += not ok
    print(f"case {case}: {m}x{l}x{n} add={add} pads {pad_a},{pad_b},{pad_c}: plan {'+'.join(f'{r}@L{x}' for r, x in plan)} vs one product at L{lv}: {'ok' if ok else 'MISMATCH'}", flush=True)
    del A, B, C, D, C0
os.environ.pop("M4RI_AMD_LEVELS", None)
print(f"ROW_BLOCKS_SOAK {'ok' if bad == 0 else 'FAILED'}: {bad} mismatches in {cases} cases, {split} of them with more than one block")
sys.exit(1 if bad else 0)
