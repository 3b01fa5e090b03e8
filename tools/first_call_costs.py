"""What the first calls of a process cost (run with PYTHONPATH=.): dlopen, m4ri_amd_init (HIP runtime + engine), the first product
(workspace, code objects), the first TRSM / PLE (their scratch).  On one MI355X: init 0.05 .. 0.13 s, first product 0.04 s, first PLE 0.01 s
-- the 0.1 .. 0.3 s that the first routine of tests/l4_timing_driver.c shows on top of its own time."""
import time, sys
t0 = time.perf_counter()
import m4ri_amd
from m4ri_amd.mzd import Mzd
t1 = time.perf_counter()
m4ri_amd.lib()
t2 = time.perf_counter()
m4ri_amd.init(0)
t3 = time.perf_counter()
A, B = Mzd.random(512, 512, 1), Mzd.random(512, 512, 2)
t4 = time.perf_counter()
m4ri_amd.mzd_mul(None, A, B, 0)
t5 = time.perf_counter()
m4ri_amd.mzd_mul(None, A, B, 0)
t6 = time.perf_counter()
U = Mzd.random(1024, 1024, 3)
m4ri_amd.mzd_trsm_upper_left(U, Mzd.random(1024, 1024, 4))
t7 = time.perf_counter()
m4ri_amd.mzd_ple(Mzd.random(1024, 1024, 5))
t8 = time.perf_counter()
m4ri_amd.mzd_ple(Mzd.random(1024, 1024, 5))
t9 = time.perf_counter()
print(f"import {t1-t0:.3f} dlopen {t2-t1:.3f} init {t3-t2:.3f} first mul {t5-t4:.3f} second mul {t6-t5:.4f} first trsm {t7-t6:.4f} first ple {t8-t7:.4f} second ple {t9-t8:.4f}")
