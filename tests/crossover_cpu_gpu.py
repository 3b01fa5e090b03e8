#!/usr/bin/env python3
"""Where does the drop-in pay off?  mzd_mul on host matrices: the reference on the host cores
(sequential SSE2 build, and mzd_mul_mp of the OpenMP build on all hardware threads) next to
libm4ri_amd.so through the same entry point (PCIe inclusive) and with the operands pinned.
Run on the GPU box (needs oracle/_ref/*.so, built by oracle/Makefile in the build container).
Lives under tests/ because it executes the reference checker (nothing outside tests/, smoke() and
bench.py's cpu_baseline leg may); it is a measurement script, not a pytest module."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))
import cpu_libs

import m4ri_amd
from m4ri_amd.mzd import Mzd


def best(fn, reps):
    fn()
    b = 1e30
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        b = min(b, time.perf_counter() - t)
    return b


m4ri_amd.init(0)
ref, omp = cpu_libs.reference(), cpu_libs.reference(openmp=True)
# ---- small products: the reference, the GPU path whatever the size (threshold 0) and the library's host routine (small_host.cpp),
#      all through mzd_mul on host matrices: where the library's size switch belongs
print(f"{'n':>6} | {'ref seq us':>10} | {'gpu path us':>11} {'host routine us':>15} | m*l*n")
old = m4ri_amd.set_small_product_threshold(0)
for n in (32, 64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 1024):
    A, B, C = Mzd.random(n, n, 3), Mzd.random(n, n, 4), Mzd.init(n, n)
    t_seq = best(lambda: ref.mul(C, A, B, 0), 20) if ref else float("nan")
    want = C.copy()
    m4ri_amd.set_small_product_threshold(0)
    t_gpu = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 20)
    assert C.equal(want)
    m4ri_amd.set_small_product_threshold(1 << 62)
    t_host = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 20)
    assert C.equal(want)
    print(f"{n:6d} | {t_seq * 1e6:10.1f} | {t_gpu * 1e6:11.1f} {t_host * 1e6:15.1f} | 2^{(3 * n.bit_length() - 3)}", flush=True)
for (m, l, n) in ((1000, 10, 20), (16, 4096, 16), (4096, 16, 64), (64, 64, 4096), (2048, 64, 64)):
    A, B, C = Mzd.random(m, l, 3), Mzd.random(l, n, 4), Mzd.init(m, n)
    t_seq = best(lambda: ref.mul(C, A, B, 0), 20) if ref else float("nan")
    m4ri_amd.set_small_product_threshold(0)
    t_gpu = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 20)
    m4ri_amd.set_small_product_threshold(1 << 62)
    t_host = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 20)
    print(f"{m}x{l}x{n}: ref {t_seq * 1e6:.1f} us | gpu path {t_gpu * 1e6:.1f} us, host routine {t_host * 1e6:.1f} us | m*l*n = {m * l * n:.2e}", flush=True)
m4ri_amd.set_small_product_threshold(old)
print(f"{'n':>6} | {'ref seq ms':>10} {'ref omp ms':>10} | {'gpu host ms':>11} {'gpu pinned ms':>13} | speedup vs best cpu (host / pinned)")
for n in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    A, B, C = Mzd.random(n, n, 3), Mzd.random(n, n, 4), Mzd.init(n, n)
    reps = 3 if n <= 8192 else 1
    t_seq = best(lambda: ref.mul(C, A, B, 0), reps) if ref else float("nan")
    t_omp = best(lambda: omp.mul_mp(C, A, B, 0), reps) if (omp and omp.has_mp) else float("nan")
    want = C.copy()
    t_gpu = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 5)
    assert C.equal(want)
    for M in (A, B, C):
        m4ri_amd.pin(M)
    t_pin = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 5)
    for M in (A, B, C):
        m4ri_amd.unpin(M)
    assert C.equal(want)
    cpu = min(t_seq, t_omp)
    print(f"{n:6d} | {t_seq * 1e3:10.3f} {t_omp * 1e3:10.3f} | {t_gpu * 1e3:11.3f} {t_pin * 1e3:13.3f} | {cpu / t_gpu:8.1f}x / {cpu / t_pin:8.1f}x", flush=True)
