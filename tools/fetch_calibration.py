#!/usr/bin/env python3
"""Which of the leaf's operands is re-read from HBM?  Run under `rocprofv3 --pmc FETCH_SIZE` (and the TCC request
counters): one product per shape, the leaf launch of each is the largest m4rm8q dispatch.  The three shapes differ in
how many row tiles share a B panel (2, 1, 4) and in the A : B byte ratio, which separates "B fetched once per row
tile" from a counter that tallies the two operands' requests differently."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

n = 65536
m4ri_amd.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
SHAPES = {"tiles_m2": (n, n, n), "tiles_m1": (n // 2, n, n), "tiles_m4": (2 * n, n, n)}
for name, (m, l, k) in SHAPES.items():
    if which not in ("all", name):
        continue
    wl, wk = l // 64, k // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wk), dtype=torch.int64, device="cuda")
    C = torch.empty((m, wk), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
    m4ri_amd.fill_dev(B.data_ptr(), wk, l, k, 4)
    m4ri_amd.set_profiling(True)
    m4ri_amd.mul_dev(C.data_ptr(), wk, A.data_ptr(), wl, B.data_ptr(), wk, m, l, k)
    torch.cuda.synchronize()
    st = m4ri_amd.get_stats()
    print(name, "levels", st.levels, "leaf", st.leaf_m, st.leaf_l, st.leaf_n, "x", st.leaf_products, "gen", st.leaf_gen, "leaf_ms", st.leaf_ms,
          "A_GB", st.leaf_products * st.leaf_m * st.leaf_l / 8e9, "B_GB", st.leaf_products * st.leaf_l * st.leaf_n / 8e9,
          "C_GB", st.leaf_products * st.leaf_m * st.leaf_n / 8e9, flush=True)
    del A, B, C
    m4ri_amd.lib().m4ri_amd_release_workspace()
