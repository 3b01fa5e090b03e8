#!/usr/bin/env python3
"""Which Strassen depth should the engine pick by itself?  For a range of shapes: the current default (leaves keep 8192 inner bits,
three-level passes) against one level more (the reference's rule with cutoff 4096: leaves of 4096) with three- and with four-level
passes.  Device-resident products, mean of `reps`, results compared."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m4ri_amd.init(0)
shapes = [(8192, 8192, 8192), (16384, 16384, 16384), (24576, 24576, 24576), (32768, 32768, 32768), (16384, 32768, 32768), (32768, 65536, 65536), (16384, 65536, 65536),
          (65536, 65536, 65536), (131072, 8192, 131072), (131072, 16384, 131072), (100003, 50021, 70017), (131072, 131072, 131072)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (m, l, n) in shapes:
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
    C = torch.empty((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
    m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
    reps = 3 if m * l * n > 2 ** 49 else 10 if m * l * n > 2 ** 44 else 30
    out, sums = [], []
    for (cutoff, fuse) in ((0, 3), (4096, 3), (4096, 4)):
        m4ri_amd.set_max_fuse(fuse)
        m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, cutoff)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, cutoff)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps * 1e3
        st = m4ri_amd.get_stats()
        out.append(f"cutoff {cutoff} fuse {fuse}: L={st.levels} leaf {st.leaf_m}x{st.leaf_l}x{st.leaf_n} {dt:8.3f} ms ws {st.workspace_bytes / 2 ** 30:5.1f} GiB")
        sums.append(int(C.sum().item()))
    print(f"{m}x{l}x{n}: " + " | ".join(out) + (" | results agree" if len(set(sums)) == 1 else " | RESULTS DIFFER"), flush=True)
    del A, B, C
    m4ri_amd.lib().m4ri_amd_release_workspace()
    torch.cuda.empty_cache()
