#!/usr/bin/env python3
"""usage: prof_product.py m l n reps [cutoff [max_fuse]]  -- `reps` device-resident products of one shape, nothing else: the workload
rocprofv3 --kernel-trace --stats is pointed at to price the Winograd passes and the leaf of that shape."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

m, l, n, reps = (int(a) for a in sys.argv[1:5])
cutoff = int(sys.argv[5]) if len(sys.argv) > 5 else 0
m4ri_amd.init(0)
if len(sys.argv) > 6:
    m4ri_amd.set_max_fuse(int(sys.argv[6]))
wl, wn = (l + 63) // 64, (n + 63) // 64
A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
B = torch.empty((l, wn), dtype=torch.int64, device="cuda")
C = torch.empty((m, wn), dtype=torch.int64, device="cuda")
m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
for _ in range(reps):
    m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, cutoff)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(reps):
    m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, cutoff)
torch.cuda.synchronize()
ms_per_product = (time.perf_counter() - t0) / max(1, reps) * 1e3
m4ri_amd.set_profiling(True)
m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, cutoff)
torch.cuda.synchronize()
st = m4ri_amd.get_stats()
print(f"shape {m}x{l}x{n}: levels {st.levels}, leaf {st.leaf_m}x{st.leaf_l}x{st.leaf_n} x{st.leaf_products}, pass bytes {st.aux_bytes / 1e9:.3f} GB, "
      f"leaf {st.leaf_ms:.3f} ms, product {ms_per_product:.3f} ms (mean of {reps}), C checksum {int(C.sum().item()) & 0xffffffffffff:012x}")
