#!/usr/bin/env python3
"""Time, on ONE GPU, the products a rank of an N-GPU run computes at n = 65536 -- the row slabs of the slabs
variant, the sub-products of the Strassen-sharded variant (x how many the busiest rank multiplies) and the blocks of
the blocks variant -- to estimate strong scaling without an 8-GPU node (compute only: the exchanges are arithmetic
in DESIGN.md 7)."""
import json
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import m4ri_amd

n = 65536
SHAPES = {"N=1 (1,1,1)": (n, n, n), "slabs N=2": (n // 2, n, n), "slabs N=4": (n // 4, n, n), "slabs N=8": (n // 8, n, n),
          "  slabs N=2, inner half (own slab of B first, then the gathered half: x2)": (n // 2, n // 2, n),
          "  slabs N=4, inner quarter (own slab / a 1-slab piece)": (n // 4, n // 4, n),
          "  slabs N=4, inner half (a 2-slab piece)": (n // 4, n // 2, n),
          "  slabs N=4, inner three quarters (a 3-slab piece)": (n // 4, 3 * n // 4, n),
          "strassen sub-product (n/2)^3 [N=8: x1]": (n // 2, n // 2, n // 2),
          "  its row half  (overlap chunks = 2: x2)": (n // 4, n // 2, n // 2),
          "  its row quarter (overlap chunks = 4: x4)": (n // 8, n // 2, n // 2),
          "  its row half x column half (overlap 2x2: x4)": (n // 4, n // 2, n // 4), "strassen sub-product (n/4)^3 [N=4: x13, N=2: x25]": (n // 4, n // 4, n // 4),
          "blocks N=4 (2,2,1)": (n // 2, n, n // 2), "blocks N=8 (4,2,1)": (n // 4, n, n // 2)}
m4ri_amd.init(0)
out = {}
for name, (m, l, k) in SHAPES.items():
    wl, wk = l // 64, k // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((l, wk), dtype=torch.int64, device="cuda")
    C = torch.empty((m, wk), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3)
    m4ri_amd.fill_dev(B.data_ptr(), wk, l, k, 4)
    for _ in range(2):
        m4ri_amd.mul_dev(C.data_ptr(), wk, A.data_ptr(), wl, B.data_ptr(), wk, m, l, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m4ri_amd.mul_dev(C.data_ptr(), wk, A.data_ptr(), wl, B.data_ptr(), wk, m, l, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    m4ri_amd.set_profiling(True)
    m4ri_amd.mul_dev(C.data_ptr(), wk, A.data_ptr(), wl, B.data_ptr(), wk, m, l, k)
    torch.cuda.synchronize()
    st = m4ri_amd.get_stats()
    m4ri_amd.set_profiling(False)
    out[name] = {"shape": [m, l, k], "ms": ms, "levels": st.levels, "leaf": [st.leaf_m, st.leaf_l, st.leaf_n],
                 "leaf_products": st.leaf_products, "leaf_gen": st.leaf_gen, "leaf_ms": st.leaf_ms,
                 "pass_GB": st.aux_bytes / 1e9}
    print(name, out[name], flush=True)
    del A, B, C
    m4ri_amd.lib().m4ri_amd_release_workspace()
base = out["N=1 (1,1,1)"]["ms"]
for name, d in out.items():
    print(f"{name}: {d['ms']:.2f} ms -> speedup {base / d['ms']:.2f}x (compute only)")


# the local passes of the Strassen-sharded variant at 8 ranks (rank 0's slabs: 1/8 of the rows of every block), and the
# piece copies a rank makes for itself -- the "passes" term of DESIGN.md 7's table, measured instead of assumed
from m4ri_amd import sharding  # noqa: E402
plan = m4ri_amd.shard_plan(8, n, n, n)
names = {"local_a": m4ri_amd.BUF_LOCAL_A, "local_b": m4ri_amd.BUF_LOCAL_B, "local_c": m4ri_amd.BUF_LOCAL_C, "child_a": m4ri_amd.BUF_CHILD_A,
         "child_b": m4ri_amd.BUF_CHILD_B, "slabs_p": m4ri_amd.BUF_SLABS_P}
bufs = {k: torch.zeros(max(1, m4ri_amd.shard_buffer_words(plan, 0, wh)), dtype=torch.int64, device="cuda") for k, wh in names.items()}
w = n // 64
stream = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
best = [1e9, 1e9]
for _ in range(6):
    ev[0].record()
    m4ri_amd.shard_down_dev(plan, 0, bufs["local_a"].data_ptr(), w, bufs["local_b"].data_ptr(), w, bufs["child_a"].data_ptr(), bufs["child_b"].data_ptr(), stream)
    ev[1].record()
    m4ri_amd.shard_up_dev(plan, 0, bufs["slabs_p"].data_ptr(), bufs["local_c"].data_ptr(), w, False, stream)
    ev[2].record()
    torch.cuda.synchronize()
    best = [min(best[0], ev[0].elapsed_time(ev[1])), min(best[1], ev[1].elapsed_time(ev[2]))]
print(f"strassen-sharded local passes at 8 ranks, n = {n}: down (A and B) {best[0]:.3f} ms, up {best[1]:.3f} ms", flush=True)
