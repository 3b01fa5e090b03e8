#!/usr/bin/env python3
"""Time of one device-resident product at EVERY Strassen depth the shape has (M4RI_AMD_LEVELS forces ONE product at that depth), next
to the plan the engine makes by itself (rows in blocks, each at its depth: engine.hip plan_row_blocks): the data the engine's time
model is checked against.  usage: depth_model_sweep.py [m,l,n ...]   (`*` = the depth of the plan's first block)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m4ri_amd.init(0)
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (12288, 12288, 12288), (16384, 16384, 16384), (20480, 20480, 20480), (24576, 24576, 24576),
          (32768, 32768, 32768), (40960, 40960, 40960), (49152, 49152, 49152), (65536, 65536, 65536),
          (32768, 4096, 32768), (65536, 4096, 65536), (65536, 8192, 65536), (131072, 8192, 131072), (131072, 16384, 131072), (131072, 4096, 131072),
          (16384, 65536, 65536), (16384, 32768, 32768), (32768, 65536, 65536), (65536, 16384, 16384), (16384, 16384, 65536), (8192, 65536, 8192),
          (8192, 8192, 131072), (131072, 8192, 8192), (262144, 8192, 32768), (100003, 50021, 70017), (50000, 12000, 90000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (m, l, n) in shapes:
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
    C = torch.empty((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
    m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
    reps = 3 if m * l * n > 2 ** 47 else 10 if m * l * n > 2 ** 43 else 30
    os.environ.pop("M4RI_AMD_LEVELS", None)
    m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, 0)
    torch.cuda.synchronize()
    auto = m4ri_amd.get_stats().levels
    for _ in range(max(2, min(20, int(0.05 / max(1e-4, m * l * n / 1e16))))):   # ~50 ms of launches: the first timed configuration of a run measured 3 - 8 % slow otherwise
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, 0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, 0)
    torch.cuda.synchronize()
    plan = m4ri_amd.plan_row_blocks(m, l, n)
    out = [f"engine's plan {'+'.join(f'{r}@L{lv}' for r, lv in plan)} {(time.perf_counter() - t) / reps * 1e3:8.3f} ms"]
    sums, seen = [int(C.sum().item())], set()
    for L in range(0, 6):
        os.environ["M4RI_AMD_LEVELS"] = str(L)
        try:
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, 0)
        except Exception as e:   # workspace beyond the device
            out.append(f"L={L}: {type(e).__name__}")
            break
        torch.cuda.synchronize()
        st = m4ri_amd.get_stats()
        if st.levels in seen:
            break
        seen.add(st.levels)
        t = time.perf_counter()
        for _ in range(reps):
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps * 1e3
        out.append(f"L={st.levels}{'*' if st.levels == auto else ''} {dt:8.3f} ms (leaf {st.leaf_m}x{st.leaf_l}x{st.leaf_n}, passes {st.aux_bytes / 1e9:.2f} GB)")
        sums.append(int(C.sum().item()))
    os.environ.pop("M4RI_AMD_LEVELS", None)
    print(f"{m}x{l}x{n}: " + " | ".join(out) + (" | results agree" if len(set(sums)) == 1 else " | RESULTS DIFFER"), flush=True)
    del A, B, C
    m4ri_amd.lib().m4ri_amd_release_workspace()
    torch.cuda.empty_cache()
