import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import m4ri_amd
from m4ri_amd.mzd import Mzd
m4ri_amd.init(0)
import numpy as np
for (m, n, sparse) in [(16384, 16384, 0), (65536, 65536, 0), (70000, 4096, 4), (30000, 30000, 5), (65536, 65536, -1), (20000, 30000, -2)]:
    A0 = Mzd.random(m, n, 1)
    for k in range(max(0, sparse)):
        A0.valid_words()[:, :] &= Mzd.random(m, n, 10 + k).valid_words()
    if sparse == -1:   # every 16th column empty: a column without a pivot in every block
        A0.valid_words()[:, :] &= np.uint64(0xFFFEFFFEFFFEFFFE)
    if sparse == -2:   # rank 3000: behind it every column is without a pivot
        A0 = m4ri_amd.mzd_mul(None, Mzd.random(m, 3000, 5), Mzd.random(3000, n, 6), 0)
    m4ri_amd.pin(A0)
    best = 1e9
    for _ in range(3):
        m4ri_amd.host_modified(A0) if hasattr(m4ri_amd, "host_modified") else None
        t = time.perf_counter(); r = m4ri_amd.mzd_ple(A0)[0]; best = min(best, time.perf_counter() - t)
    m4ri_amd.unpin(A0)
    print(f"ple {m}x{n} sparse={sparse}: {best*1e3:.1f} ms rank {r}", flush=True)
