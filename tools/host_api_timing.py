#!/usr/bin/env python3
"""API-to-API timing of the drop-in host entry point (mzd_mul on host mzd_t matrices: H2D of A and B,
device schedule, D2H of C) next to the device-resident time -- the PCIe-inclusive rate quoted in
DESIGN.md.  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m4ri_amd
from m4ri_amd.mzd import Mzd

m4ri_amd.init(0)
for n in (4096, 16384, 32768, 65536):
    A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
    C = Mzd.init(n, n)
    m4ri_amd.mzd_mul(C, A, B, 0)  # warm-up (workspace allocation, code load)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        m4ri_amd.mzd_mul(C, A, B, 0)
        best = min(best, time.perf_counter() - t)
    gib = 3 * n * n / 8 / 2 ** 30
    print(f"mzd_mul host API n={n}: {best * 1e3:9.2f} ms  -> {n ** 3 / best:.3e} bit-op/s  ({gib:.2f} GiB over PCIe, "
          f"{gib / best:.1f} GiB/s if it were all transfer)", flush=True)
    old = m4ri_amd.set_host_pipeline(0)   # the one-shot schedule: upload everything, multiply, download
    m4ri_amd.mzd_mul(C, A, B, 0)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        m4ri_amd.mzd_mul(C, A, B, 0)
        best = min(best, time.perf_counter() - t)
    m4ri_amd.set_host_pipeline(old)
    print(f"mzd_mul host API n={n}, slab pipeline off: {best * 1e3:9.2f} ms", flush=True)
    # the same call with all three matrices pinned (include/m4ri_amd.h part 3): nothing crosses PCIe
    for M in (A, B, C):
        m4ri_amd.pin(M)
    m4ri_amd.mzd_mul(C, A, B, 0)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        m4ri_amd.mzd_mul(C, A, B, 0)
        best = min(best, time.perf_counter() - t)
    for M in (A, B, C):
        m4ri_amd.unpin(M)
    print(f"mzd_mul host API n={n}, A/B/C pinned: {best * 1e3:9.2f} ms  -> {n ** 3 / best:.3e} bit-op/s", flush=True)
