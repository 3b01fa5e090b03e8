#!/usr/bin/env python3
"""131072^3 on one GPU (4 Strassen levels, 2401 leaves, ~70 GiB of workspace): two different schedules
must agree bit for bit.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, m4ri_amd
m4ri_amd.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072; w = n // 64
A = torch.empty((n, w), dtype=torch.int64, device="cuda"); B = torch.empty_like(A)
m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 3); m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 4)
outs = []
for cutoff, fuse in ((0, 3), (n // 8, 2)):
    m4ri_amd.set_max_fuse(fuse)
    C = torch.empty_like(A)
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, cutoff=cutoff)
    torch.cuda.synchronize()
    t = time.perf_counter()
    m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, cutoff=cutoff)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = m4ri_amd.get_stats()
    print(f"{n}^3 cutoff={cutoff} fuse={fuse}: levels {st.levels}, leaf {st.leaf_m}x{st.leaf_l}x{st.leaf_n} x{st.leaf_products}, {dt*1e3:.1f} ms, {n**3/dt:.3e} bit-op/s, ws {st.workspace_bytes/2**30:.1f} GiB", flush=True)
    outs.append(C)
print("schedules agree:", bool(torch.equal(outs[0], outs[1])))
