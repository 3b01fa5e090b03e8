#!/usr/bin/env python3
"""usage: passes_only.py n reps  -- the fused two-level Winograd passes of an n x n operand (down on A and B, up on C) `reps` times
and nothing else: the HBM-bound half of the schedule as a workload of its own (power traces, rocprofv3).  Uses the sharding
plan at world size 1, whose local passes are exactly the engine's down2 / up2 kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

n, reps = int(sys.argv[1]), int(sys.argv[2])
m4ri_amd.init(0)
plan = m4ri_amd.shard_plan(1, n, n, n, 2)
w = n // 64
buf = {k: torch.empty(max(1, m4ri_amd.shard_buffer_words(plan, 0, wh)), dtype=torch.int64, device="cuda")
       for k, wh in (("la", m4ri_amd.BUF_LOCAL_A), ("lb", m4ri_amd.BUF_LOCAL_B), ("lc", m4ri_amd.BUF_LOCAL_C), ("ca", m4ri_amd.BUF_CHILD_A),
                     ("cb", m4ri_amd.BUF_CHILD_B), ("sp", m4ri_amd.BUF_SLABS_P))}
m4ri_amd.fill_dev(buf["la"].data_ptr(), w, n, n, 3)
m4ri_amd.fill_dev(buf["lb"].data_ptr(), w, n, n, 4)
buf["sp"].zero_()


def once():
    m4ri_amd.shard_down_dev(plan, 0, buf["la"].data_ptr(), w, buf["lb"].data_ptr(), w, buf["ca"].data_ptr(), buf["cb"].data_ptr())
    m4ri_amd.shard_up_dev(plan, 0, buf["sp"].data_ptr(), buf["lc"].data_ptr(), w, False)


once()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
gb = 8.0 * 65 * (n // 4) * (n // 4 // 64) * 3 / 1e9   # 16 blocks in + 49 children out, three operands
print(f"passes n={n}: {dt * 1e3:.3f} ms per (down2 A, down2 B, up2 C) = {gb / dt / 1e3:.2f} TB/s over {gb:.2f} GB")
