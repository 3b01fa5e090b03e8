"""Resident mzd_trsm_upper_left at a few (rows, columns) shapes, best of 5 -- run with M4RI_AMD_TRSM_BIG=0 / 1 to compare the
512-row and the 4096-row block inverses (trsm.hip)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m4ri_amd
from m4ri_amd.mzd import Mzd

m4ri_amd.init(0)
for mb, nb in ((4608, 4608), (6144, 6144), (8192, 512), (8192, 2048), (8192, 8192), (16384, 1024), (16384, 4096), (32768, 2048)):
    T, B = Mzd.random(mb, mb, 3), Mzd.random(mb, nb, 2)
    m4ri_amd.pin(T); m4ri_amd.pin(B)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); m4ri_amd.mzd_trsm_upper_left(T, B); ts.append(time.perf_counter() - t)
    m4ri_amd.unpin(T); m4ri_amd.unpin(B)
    print(f"BIG={os.environ.get('M4RI_AMD_TRSM_BIG', 'default')}: {mb:6d} x {nb:6d}: {min(ts[1:]) * 1e3:8.2f} ms", flush=True)
