"""bench.py's `cpu_baseline` leg: the real reference M4RI (oracle/_ref, built from /root/reference by oracle/Makefile) -- or, where that
binary is absent, the oracle's plain-C restatement -- timed on THIS host's cores.  Test infrastructure used as a reported baseline
only: nothing here is imported by the product (m4ri_amd/), and bench.py calls it outside every timed GPU region."""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sysfs_cache_sizes():
    """L1/L2/L3 the way the reference's configure reads them (m4/ax_cache_size.m4:46-58): for index 0..3 of
    cpu0, L<level> = size (a later index of the same level overwrites an earlier one)."""
    out = {}
    for idx in range(4):
        base = f"/sys/devices/system/cpu/cpu0/cache/index{idx}"
        try:
            level = int(open(base + "/level").read())
            size = open(base + "/size").read().strip()
        except OSError:
            continue
        mult = {"K": 1024, "M": 1 << 20, "G": 1 << 30}.get(size[-1].upper(), 1)
        out[level] = int(size.rstrip("KMGkmg")) * mult
    return out.get(1), out.get(2), out.get(3)


def cpu_baseline(n_workload: int, budget_s: float = 75.0):
    import ctypes
    import cpu_libs
    from m4ri_amd.mzd import Mzd, MzdPtr
    ncpu = os.cpu_count() or 1
    l1, l2, l3 = sysfs_cache_sizes()
    tag = f"_c{l1}_{l2}_{l3}" if l1 and l2 and l3 else ""
    ref = cpu_libs.reference(tag=tag) or cpu_libs.reference()
    matched = cpu_libs.reference(tag=tag) is not None and bool(tag)
    cache_note = (f"cache macros = this host's sysfs values L1/L2/L3 = {l1}/{l2}/{l3}" if matched else
                  f"cache macros 32768/2097152/33554432 (no build for this host's sysfs values {l1}/{l2}/{l3} in oracle/_ref)")
    if ref is None:
        orc = cpu_libs.oracle()  # no reference binary on this box: time our own plain-C restatement instead
        n = 4096
        A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
        t = time.perf_counter()
        orc.mul(None, A, B, 0)
        dt = time.perf_counter() - t
        return {"value": n ** 3 / dt, "unit": "bit-op/s", "cores": 1, "kind": "port", "sample": f"oracle gf2o_mul {n}^3, 1 run, {dt:.2f} s"}
    out = {"unit": "bit-op/s", "kind": "reference", "cache": cache_note}
    # (1) BASELINE.json configs[0]: bench_multiplication 4096 4096 4096 -- srandom(17), mzd_randomize'd A and B,
    #     timed region = mzd_mul(NULL, A, B, 0) including the allocation of C (bench/bench_multiplication.c:86-107)
    libc = ctypes.CDLL(None)
    libc.srandom(17)
    rnd = ref.L.mzd_randomize
    rnd.restype, rnd.argtypes = None, [MzdPtr]
    A1, B1 = Mzd.init(4096, 4096), Mzd.init(4096, 4096)
    rnd(A1.ptr)
    rnd(B1.ptr)
    # the reference's own stop rule (bench/benchmarking.c:502-603 with its defaults, benchmarking.c:81-91): at least 2 samples, at most
    # 1000, until the 99 % confidence interval of the mean (Student's t) is within 1 % of the mean or 60 s have passed -- on wall time
    # here (`-s 0`), capped at 10 s so that the default run stays short; one untimed warm-up first
    from scipy import stats as _st
    r = ref.L.mzd_mul(None, A1.ptr, B1.ptr, 0)
    ref.L.mzd_free(r)
    ts, t_start, ci_rel = [], time.perf_counter(), None
    while len(ts) < 1000:
        t = time.perf_counter()
        r = ref.L.mzd_mul(None, A1.ptr, B1.ptr, 0)
        ts.append(time.perf_counter() - t)
        ref.L.mzd_free(r)
        if len(ts) >= 2:
            mean = sum(ts) / len(ts)
            sd = (sum((x - mean) ** 2 for x in ts) / (len(ts) - 1)) ** 0.5
            ci_rel = float(_st.t.ppf(0.995, len(ts) - 1)) * sd / len(ts) ** 0.5 / mean
            if ci_rel <= 0.01 or time.perf_counter() - t_start > 10.0:
                break
    out["config1"] = {"what": "bench_multiplication 4096 4096 4096: mzd_mul(NULL,A,B,0) incl. allocating C, srandom(17) + mzd_randomize inputs, "
                              "sequential SSE2 build; the reference's stop rule (bench/benchmarking.c:502-603): >= 2 samples until the 99 % "
                              "confidence interval of the mean is within 1 % of it, on wall time, at most 1000 samples / 10 s",
                      "samples": len(ts), "ci99_rel": ci_rel,
                      "seconds_mean": sum(ts) / len(ts), "seconds_min": min(ts), "bitops_per_sec": 4096 ** 3 / (sum(ts) / len(ts)), "cores": 1}
    # (2) the workload itself on all cores: mzd_mul_mp (OpenMP build), once, if a 16384^3 probe says it fits the budget
    omp = cpu_libs.reference(openmp=True, tag=tag) or cpu_libs.reference(openmp=True)
    n = 16384
    A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
    t = time.perf_counter()
    ref.mul(None, A, B, 0)
    t_seq = time.perf_counter() - t
    out["sequential"] = {"value": n ** 3 / t_seq, "cores": 1, "sample": f"mzd_mul {n}^3, sequential build, 1 run: {t_seq:.2f} s"}
    out.update({"value": n ** 3 / t_seq, "cores": 1, "sample": out["sequential"]["sample"]})
    if omp is not None and omp.has_mp:
        os.environ.setdefault("OMP_NUM_THREADS", str(ncpu))
        best = 1e30
        for _ in range(2):
            t = time.perf_counter()
            omp.mul_mp(None, A, B, 0)
            best = min(best, time.perf_counter() - t)
        out["openmp_16384"] = {"value": n ** 3 / best, "cores": ncpu, "sample": f"mzd_mul_mp {n}^3, OpenMP build, {ncpu} threads, best of 2: {best:.2f} s"}
        out.update({"value": n ** 3 / best, "cores": ncpu, "sample": out["openmp_16384"]["sample"]})
        # BASELINE.md 3 asks for both calls on all cores: the OpenMP build's plain mzd_mul (row-parallel M4RM leaves,
        # brilliantrussian.c:1121-1123, sequential Strassen) beside mzd_mul_mp (2 x 2 blocks of C, mp.c:206-228).  It forks and
        # joins a team per table step and gets SLOWER with cores (16384^3 on 256 threads: 19.7 s against 0.92 s sequential,
        # profiles/r03_bench65536_first.json), so the sample is one 8192^3 product
        n8 = 8192
        A8, B8 = Mzd.random(n8, n8, 3), Mzd.random(n8, n8, 4)
        t = time.perf_counter()
        omp.mul(None, A8, B8, 0)
        t_mul = time.perf_counter() - t
        out["openmp_mzd_mul_8192"] = {"value": n8 ** 3 / t_mul, "cores": ncpu,
                                      "sample": f"mzd_mul {n8}^3, OpenMP build, {ncpu} threads, 1 run: {t_mul:.2f} s"}
        del A8, B8
        predicted = best * (n_workload / n) ** 2.807
        if n_workload > n and predicted <= budget_s:
            del A, B
            A, B = Mzd.random(n_workload, n_workload, 3), Mzd.random(n_workload, n_workload, 4)
            t = time.perf_counter()
            omp.mul_mp(None, A, B, 0)
            dt = time.perf_counter() - t
            out.update({"value": n_workload ** 3 / dt, "cores": ncpu,
                        "sample": f"the workload itself: reference mzd_mul_mp {n_workload}^3 (same splitmix64 inputs), OpenMP build, "
                                  f"OMP_NUM_THREADS={ncpu}, 1 run: {dt:.2f} s"})
        elif n_workload > n:
            out["sample"] += f"; the {n_workload}^3 run was skipped (predicted {predicted:.0f} s > budget {budget_s:.0f} s)"
    return out
