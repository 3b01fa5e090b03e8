#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X:

    GF(2) n^3-equivalent bit-ops/s and wall-clock of one n x n x n mzd_mul, n = 65536,
    at 1/2/4/8 GPUs (strong scaling: the total work is fixed).

A "step" is one whole product C = A*B on device-resident, synthetic (splitmix64, density 1/2) operands,
everything through libm4ri_amd.so's C ABI.  Inputs are in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 65536] [--workload mul|leaf16384|rect131072]

N = 1: Strassen-Winograd levels over batched M4RM leaves on one GPU (m4ri_amd_mul_dev).

N > 1 (one process per GPU, torch.distributed.run, RCCL); --variant auto (default) = slabs up to 4 ranks, strassen above:
  --variant slabs: rank r holds rows [r m/N, (r+1) m/N) of A, of B and of C; ONE RCCL collective, the all-gather of
      B's row slabs (all_gather_into_tensor), then every rank multiplies its slab of A by the whole B.  Gives up
      log2(N) Strassen levels in the row direction, which is cheap at 2 and 4 ranks (and 2 ranks share one xGMI
      link, which a Strassen-sharded exchange would saturate).
  --variant strassen: the sub-products of the top Strassen-Winograd level(s) are spread over the
      ranks (m4ri_amd/csrc/multi.hip, m4ri_amd/sharding.py).  A, B and C are distributed slab-cyclically
      (rank r holds rows [cut(r), cut(r+1)) of every row block): every rank runs the level's additions on
      its own slabs, slabs of sub-product operands travel rank -> owner and slabs of products back, each as
      one RCCL send/recv on the direct xGMI link of its pair, all posted as one batch per phase.
      --layout distributed (default): the slabs ARE where the inputs live when the timed region starts and
          where C is left (the layout products chain in);
      --layout owner: A and B live on rank 0 and C is gathered there: scatter and gather are inside the
          timed region (bounded by rank 0's seven links: documented in DESIGN.md 7).
  --variant blocks: the reference's own template (_mzd_mul_mp4, m4ri/mp.c:158-275): a grid of blocks of C,
      optionally the inner dimension split with ONE pairwise XOR exchange (--grid 2,2,2); blocks of A and B
      are scattered from rank 0 and the reduced blocks of C gathered there (owner layout only).

`value` / `ms_per_step` are always ONE product at a time (the metric is the wall clock of one n x n mzd_mul).  --inflight 2 adds a second,
separately named measurement: `pipelined_value` / `pipelined_ms_per_step`, the throughput of a stream of independent products with two
in flight (sharding.run_products: the transport of the neighbouring products runs under the multiplications of the current one).

N > 1 transports (--transport): `rccl` = one process per GPU, torch.distributed (backend nccl == RCCL) send/recv + all-gather, as above;
`peer` = ONE process driving all GPUs through libm4ri_amd.so's distributed matrices (m4ri_amd_dmat_mul, m4ri_amd/csrc/multi.hip: the same
two schedules behind the C boundary, pieces pulled by hipMemcpyPeerAsync on copy streams, one host thread per GPU).  The command that
receives `--gpus N` is a CONTROLLER: it starts the ranks itself, watches them (--watchdog seconds) and walks down a ladder
rccl -> peer -> a JSON error line, so that an N-GPU run always ends with exactly one line (`config.transport`, `config.transport_fallback`).
Under `torch.distributed.run` rank 0 is the controller and the other launcher ranks step aside.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline"     the dominant kernel (the M4RM leaf launch) against the HBM roofline: duration = mean over ALL
                 timed steps of HIP events around that launch on its stream; "traffic" = HBM bytes per launch
                 from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) this script runs on itself (N = 1);
  "cpu_baseline" the real reference M4RI (oracle/_ref, built from /root/reference) timed on this host's
                 cores: mzd_mul_mp at the workload's own size, and the reference's bench_multiplication
                 4096^3 timed region (BASELINE.json configs[0]);
  "verified"     SHA-256 of the C the timed steps produced against the reference's (tests/golden).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the reference itself on this host's cores
# ---------------------------------------------------------------------------------------------------
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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import cpu_libs
    from m4ri_amd.mzd import Mzd, MzdPtr, from_struct_ptr
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


# ---------------------------------------------------------------------------------------------------
# models and measurements around the number
# ---------------------------------------------------------------------------------------------------
LEAF_KERNELS = {1: "m4rm_leaf_kernel", 3: "m4rm8_kernel", 4: "m4rm8q_kernel"}
# (tile rows, tile columns, inner bits per stage, LDS-array clocks per stage): gathers at 256 B/clk/CU
# + table writes at 128 B/clk/CU, both measured with tools/ubench.hip
LEAF_LDS_MODEL = {3: (2048, 1024, 16, 2560), 4: (4096, 512, 32, 4608)}
CU_COUNT, PEAK_CLOCK_HZ = 256, 2.4e9


def lds_model(gen, m, l, n, products, launch_ms):
    """The leaf launch against the LDS-array bound of its own design: CU-clocks the gathers and table
    writes need at the measured LDS rates, spread over 256 CUs at the 2.4 GHz peak clock."""
    if gen not in LEAF_LDS_MODEL or launch_ms <= 0:
        return None
    tr, tc, bits, clk = LEAF_LDS_MODEL[gen]
    tiles = -(-m // tr) * -(-n // tc)
    stages = -(-l // bits)
    bound_ms = products * tiles * stages * clk / (CU_COUNT * PEAK_CLOCK_HZ) * 1e3
    return {"bound_ms": bound_ms, "frac": bound_ms / launch_ms, "clk_per_stage": clk,
            "tile": [tr, tc], "bits_per_stage": bits, "clock_hz": PEAK_CLOCK_HZ}


def measured_copy_gbs():
    """On-box HBM copy rate (GB/s, read + write bytes) of a 1 GiB device-to-device copy: the practical
    peak SURVEY.md 8(d) asks to report beside the vendor 8 TB/s."""
    src = torch.empty(1 << 27, dtype=torch.int64, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return 2.0 * src.numel() * 8 / (best * 1e-3) / 1e9


def measure_leaf_traffic(argv_size, cutoff, timeout_s=240):
    """HBM bytes of ONE leaf launch of this workload from rocprofv3 PMC passes run right now, on this box,
    over this script in probe mode (one warm-up + one product): FETCH_SIZE and WRITE_SIZE in separate
    passes (they do not fit one), per-dispatch sums over all instances, bytes = (2*FETCH_SIZE + WRITE_SIZE)
    * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read
    stream), WRITE_SIZE as reported.  None (with the reason) when rocprofv3 is unavailable or fails."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="m4ri_amd_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "GRBM_GUI_ACTIVE", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--probe",
               "--size", str(argv_size), "--cutoff", str(cutoff)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("results.db")]
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-300:]}"
            db = sqlite3.connect(dbs[0])
            names = dict(db.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
            disp = {ev: (names.get(kid, ""), end - start) for kid, ev, start, end in
                    db.execute("select kernel_id, event_id, start, end from rocpd_kernel_dispatch")}
            pmc = {pid: nm for pid, nm in db.execute("select id, name from rocpd_info_pmc")}
            per_dispatch, cycles = {}, {}
            for ev, pid, val in db.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
                if "m4rm" not in disp.get(ev, ("", 0))[0]:
                    continue
                if pmc.get(pid) == counter:
                    per_dispatch[ev] = per_dispatch.get(ev, 0.0) + val
                elif pmc.get(pid) == "GRBM_GUI_ACTIVE":
                    cycles[ev] = max(cycles.get(ev, 0.0), val)
            if not per_dispatch:
                return None, f"no {counter} rows for a leaf kernel"
            ev = max(per_dispatch, key=per_dispatch.get)  # the batched leaf launch (strips, if any, are smaller)
            vals[counter] = per_dispatch[ev]
            if ev in cycles and disp[ev][1] > 0:
                vals["gui_cycles"], vals["profiled_ns"] = cycles[ev], disp[ev][1]
        except Exception as e:  # noqa: BLE001
            return None, f"{counter}: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    detail = {"fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"]}
    if "gui_cycles" in vals:  # the clock the launch really ran at (under the profiler): GRBM_GUI_ACTIVE / duration
        detail.update({"gui_active_cycles": vals["gui_cycles"], "profiled_launch_ms": vals["profiled_ns"] * 1e-6,
                       "effective_clock_hz": vals["gui_cycles"] / (vals["profiled_ns"] * 1e-9)})
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, detail


def host_api_timing(A_dev, B_dev, M, L, N, cutoff, runs=3):
    """The other timed region SURVEY.md 8(d) asks for: entry to return of the drop-in host entry point on ordinary
    (pageable) host mzd_t matrices -- H2D of A and B, the device schedule, D2H of C -- next to the device-resident
    number.  Two forms: C given (allocated and touched beforehand) and C == NULL, which is what the reference's
    bench times (bench/bench_multiplication.c:85-107: mzd_mul(NULL, A, B, cutoff) including the allocation of C; here
    the fresh 64-byte-aligned block's pages are first touched by the download).  min / median of `runs` after one warm-up."""
    import ctypes
    from m4ri_amd.mzd import Mzd
    lib = m4ri_amd.lib()
    Ah, Bh, Ch = Mzd(M, L), Mzd(L, N), Mzd(M, N)
    Ah.valid_words()[:, :] = A_dev.cpu().numpy().view(np.uint64)   # the very operands of the timed steps, now in host memory
    Bh.valid_words()[:, :] = B_dev.cpu().numpy().view(np.uint64)
    Ch.buf.fill(0)                                                  # touch C's pages: "C given" means a matrix the caller already uses
    out = {}
    lib.mzd_mul(Ch.ptr, Ah.ptr, Bh.ptr, cutoff)                     # warm-up: staging arena, host pipeline threads
    ts = []
    for _ in range(runs):
        t = time.perf_counter()
        lib.mzd_mul(Ch.ptr, Ah.ptr, Bh.ptr, cutoff)
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    out["c_given_ms_min"], out["c_given_ms_median"] = ts[0], ts[len(ts) // 2]
    ts = []
    for _ in range(runs):
        t = time.perf_counter()
        r = lib.mzd_mul(None, Ah.ptr, Bh.ptr, cutoff)
        ts.append((time.perf_counter() - t) * 1e3)
        lib.m4ri_amd_result_free(r)
    ts.sort()
    out["c_null_ms_min"], out["c_null_ms_median"] = ts[0], ts[len(ts) // 2]
    gib = 8.0 * (M * Ah.width + L * Bh.width + M * Ch.width) / 2 ** 30
    out.update({"runs": runs, "GiB_over_pcie": gib, "bitops_per_sec": float(M) * L * N / (out["c_given_ms_min"] * 1e-3),
                "what": "host mzd_mul(C, A, B, cutoff) on pageable mzd_t matrices, PCIe transfers (and for c_null the allocation of C) "
                        "inside the timed region; never the headline `value`"})
    return out


def bytes_sched(m, l, n, levels):
    """HBM bytes of the DECLARED, UNFUSED Strassen-Winograd schedule for C = A*B with
    `levels` levels -- every operand read once and every result written once per kernel: a leaf `mul`
    moves 8*(m*W(l) + l*W(n) + m*W(n)) bytes, a quadrant addition of r x c bits 3*8*r*W(c); per level 7
    products + 15 additions (4 on A-quadrant shape, 4 on B-quadrant shape, 7 on C-quadrant shape,
    strassen.c:111-150)."""
    W = lambda x: (x + 63) // 64  # noqa: E731
    if levels == 0:
        return 8 * (m * W(l) + l * W(n) + m * W(n))
    hm, hl, hn = m // 2, l // 2, n // 2
    adds = 3 * 8 * (4 * hm * W(hl) + 4 * hl * W(hn) + 7 * hm * W(hn))
    return adds + 7 * bytes_sched(hm, hl, hn, levels - 1)


def golden_sha(op, m, l, n, seeds):
    """SHA-256 of the reference's result for this product, if tests/golden holds one (make_golden.py --sha)."""
    p = os.path.join(ROOT, "tests", "golden", "sha256.json")
    if not os.path.exists(p):
        return None
    for e in json.load(open(p)):
        if (e["op"], e["m"], e["l"], e["n"], e["seed_a"], e["seed_b"]) == (op, m, l, n, seeds[0], seeds[1]) and not e.get("cutoff"):
            return e["sha256"]
    return None


def sha_of_device_rows(t, chunk_rows=8192):
    """SHA-256 over the words of a 2-D int64 device tensor, row-major (== over the valid bits when the
    width is whole words), streamed through the host in chunks."""
    h = hashlib.sha256()
    for r0 in range(0, t.shape[0], chunk_rows):
        h.update(t[r0:r0 + chunk_rows].cpu().numpy().tobytes())
    return h.hexdigest()


LAUNCHER_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                "MASTER_ADDR", "MASTER_PORT", "NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_ASYNC_ERROR_HANDLING")


def run_rung(transport: str, n_ranks: int, argv: list, watchdog_s: float):
    """One rung of the N > 1 ladder in processes of its own (own session: on a timeout the whole group we started is killed, nothing
    else).  rccl: `torch.distributed.run --standalone` (it picks its own rendezvous port on 127.0.0.1) with one rank per GPU; peer: one
    process driving all GPUs.  Returns (the rung's JSON line or None, why it failed or None, its other stdout lines)."""
    import signal
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV and not k.startswith("TORCHELASTIC_")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    args, skip = [], False
    for a in argv:  # the rung gets this command's own arguments, minus what the controller decides
        if skip:
            skip = False
        elif a in ("--transport", "--watchdog"):
            skip = True
        elif not (a.startswith("--transport=") or a.startswith("--watchdog=") or a == "--inner"):
            args.append(a)
    me = os.path.abspath(__file__)
    if transport == "rccl":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n_ranks}",
               me, *args, "--inner", "--transport", "rccl"]
    else:
        cmd = [sys.executable, me, *args, "--inner", "--transport", "peer"]
    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:
        proc = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, start_new_session=True)
        why = None
        try:
            rc = proc.wait(timeout=watchdog_s)
            if rc != 0:
                why = f"exit code {rc}"
        except subprocess.TimeoutExpired:
            why = f"no result within the {watchdog_s:.0f} s watchdog: ranks killed"
        finally:
            try:  # the session we started (torchrun and its ranks), nothing else; a finished group is simply gone
                os.killpg(proc.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            proc.wait()
        fo.seek(0)
        fe.seek(0)
        out_lines, err_tail = fo.read().splitlines(), fe.read()[-1500:]
    lines = [ln for ln in out_lines if ln.startswith("{") and '"metric"' in ln]
    rest = [ln for ln in out_lines if not (ln.startswith("{") and '"metric"' in ln)]
    if why is None and len(lines) != 1:
        why = f"{len(lines)} result lines"
    if why is not None:
        errs = [ln for ln in out_lines if ln.startswith("{") and '"error"' in ln]
        detail = errs[-1] if errs else " | ".join(err_tail.strip().splitlines()[-3:])
        return None, f"{why}: {detail}"[:700], rest
    return lines[0], None, rest


def controller(args) -> int:
    """`bench.py --gpus N` as a command: start the ranks, watch them, fall back down the ladder rccl -> peer, and end with exactly
    one JSON line either way (the reference switches to its multi-core path inside the same command, bench/bench_multiplication.c:94-103).
    With `--transport auto` on the real backend BOTH transports are measured when both work -- the same product through two complete
    implementations -- and the line is the faster one's, the other's numbers beside it (`config.transports_measured`)."""
    ladder = {"auto": ["rccl", "peer"], "rccl": ["rccl"], "peer": ["peer"]}[args.transport]
    if args.variant == "blocks" or args.layout == "owner":
        ladder = [t for t in ladder if t == "rccl"] or ["rccl"]  # scatter / gather layouts exist over torch.distributed only
    measure_all = args.transport == "auto" and args.backend == "nccl" and len(ladder) > 1
    failed, done = [], {}
    for transport in ladder:
        line, why, rest = run_rung(transport, args.gpus, sys.argv[1:], args.watchdog)
        for ln in rest:
            print(ln, flush=True)
        if line is not None:
            done[transport] = json.loads(line)
            if not measure_all:
                break
        else:
            failed.append({"transport": transport, "reason": why})
    if not done:
        print(json.dumps({"error": f"no transport completed the {args.gpus}-GPU run", "gpus_requested": args.gpus, "transport_fallback": failed}), flush=True)
        return 1
    best = min(done, key=lambda t: done[t]["ms_per_step"])
    out = done[best]
    out["config"]["transport"] = best
    first_ok = min(ladder.index(t) for t in done)
    out["config"]["transport_fallback"] = [f for f in failed if ladder.index(f["transport"]) < first_ok]
    later = [f for f in failed if ladder.index(f["transport"]) > first_ok]
    if later:
        out["config"]["transports_unavailable"] = later
    if len(done) > 1:
        out["config"]["transports_measured"] = {t: {"value": d["value"], "ms_per_step": d["ms_per_step"], "host_issue_ms_per_step": d.get("host_issue_ms_per_step")}
                                                for t, d in done.items()}
    print(json.dumps(out), flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=65536, help="n of the n x n x n product")
    ap.add_argument("--workload", default="mul", choices=["mul", "leaf16384", "rect131072"],
                    help="mul: n^3 mzd_mul (the headline, configs[2]/[3]); leaf16384: configs[1]; "
                         "rect131072: 131072 x 8192 x 131072 (configs[4])")
    ap.add_argument("--cutoff", type=int, default=0)
    ap.add_argument("--variant", default="auto", choices=["auto", "strassen", "slabs", "blocks"],
                    help="N > 1: what is handed out to the ranks (auto: slabs up to 4 ranks, strassen above)")
    ap.add_argument("--layout", default="distributed", choices=["distributed", "owner"], help="N > 1: where A, B live and C is left")
    ap.add_argument("--shard-levels", type=int, default=0, help="strassen variant: sharded levels (1, 2; 0 = automatic)")
    ap.add_argument("--grid", default="", help="blocks variant: gi,gj,gh split of (m, n, l) over the ranks (default: sharding.default_grid)")
    ap.add_argument("--max-fuse", type=int, default=0, help="Strassen levels per fused pass (1..4; 0 = engine default: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the host-API (PCIe-inclusive) timing")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--no-verify", action="store_true", help="skip the SHA-256 check of C against the reference's")
    ap.add_argument("--probe", action="store_true", help="internal: one warm-up + one product, nothing else (run under rocprofv3)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: development aid -- several ranks may share one GPU, pieces are staged through the host")
    ap.add_argument("--check", action="store_true",
                    help="after timing, every rank recomputes the full product on its own GPU and compares the part of C it holds")
    ap.add_argument("--seeds", default="", help="a,b: splitmix64 seeds of A and B (default 3,4; 5,6 for rect131072)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: initialise the process group anyway and run the N > 1 code path (its collectives, "
                         "batches and fences) at world size 1 -- how a one-GPU box puts the RCCL path through its paces")
    ap.add_argument("--inflight", type=int, default=1, choices=[1, 2],
                    help="N > 1, distributed layout, rccl transport: 2 = after the timed loop (always one product at a time: `value`) also time a "
                         "software-pipelined stream of products, two in flight -- the transport of product k+1 (operands) and of product k-1 (results) "
                         "under the multiplications of product k -- reported as pipelined_value / pipelined_ms_per_step")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "peer"],
                    help="N > 1: rccl = one process per GPU over torch.distributed; peer = one process, all GPUs, m4ri_amd_dmat_mul (peer copies); "
                         "auto = rccl, falling back to peer when the ranks fail or do not finish within --watchdog seconds")
    ap.add_argument("--watchdog", type=float, default=240.0, help="N > 1: seconds one rung of the transport ladder may take before its processes are killed")
    ap.add_argument("--virtual-ranks", action="store_true",
                    help="peer transport: allow more ranks than visible GPUs (ranks share devices round robin: how a one-GPU box tests the path)")
    ap.add_argument("--inner", action="store_true", help="internal: this process is a rank / the single process of a rung the controller started")
    ap.add_argument("--slab-overlap", type=int, default=-1,
                    help="slabs variant: multiply with the rank's own slab of B while the all-gather of the others runs "
                         "(1 / 0; default: on up to 4 ranks)")
    ap.add_argument("--dims", default="", help="m,l,n of a general product (overrides --size; ragged sizes exercise the uneven slabs)")
    ap.add_argument("--overlap", default="",
                    help="strassen variant: R or RxC -- row (x column) chunks per sub-product whose transport overlaps the products "
                         "(1 = none; default: 2 when a rank owns one large sub-product, else 1)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it (the reference switches to its multi-core path inside the
    # same command, bench/bench_multiplication.c:94-103): start the N ranks ourselves, one per GPU, and let rank 0 print
    if args.gpus > 1 and not args.inner and not args.probe:
        # this command is the controller of an N-GPU run.  Under a launcher that already started N copies of it, copy 0 takes the
        # role and the others step aside: the ranks that do the work are the controller's own, watched and replaceable
        if int(os.environ.get("RANK", "0")) != 0:
            return
        raise SystemExit(controller(args))
    peer = args.transport == "peer" and args.gpus > 1
    world = args.gpus if peer else int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0 if peer else int(os.environ.get("RANK", "0"))
    local_rank = 0 if peer else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never print an n_gpus the command did not ask for
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    multi = world > 1 or args.force_dist   # the distributed code path (normally N > 1)
    if peer:
        ndev = m4ri_amd.lib().m4ri_amd_device_count()
        if ndev < world and not args.virtual_ranks:
            print(json.dumps({"error": f"--transport peer --gpus {world}: only {ndev} device(s) visible (--virtual-ranks lets ranks share devices: a test aid)"}), flush=True)
            raise SystemExit(2)
        m4ri_amd.set_devices([i % max(1, ndev) for i in range(world)])
        torch.cuda.set_device(0)
    elif multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        solo = world == 1 and "MASTER_PORT" not in os.environ   # --force-dist typed by hand: a store on a port of its own choosing
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if args.variant == "auto":
                args.variant = "strassen"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        import datetime
        pg_timeout = datetime.timedelta(seconds=max(60.0, min(args.watchdog, 600.0)))   # a rank that never arrives is an error, not a hang
        rdzv = {"init_method": "tcp://127.0.0.1:0", "rank": 0, "world_size": 1} if solo else {}
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout, **rdzv)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout, **rdzv)
        inject = os.environ.get("M4RI_AMD_BENCH_INJECT", "") if args.inner else ""   # test hook: the first rung fails / never finishes
        if inject == "crash":
            raise SystemExit(9)
        if inject == "hang":
            time.sleep(1e6)
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    m4ri_amd.init(dev)
    if args.max_fuse:
        m4ri_amd.set_max_fuse(args.max_fuse)
    stream = torch.cuda.current_stream().cuda_stream

    n = 16384 if args.workload == "leaf16384" else args.size
    M, L, N = (131072, 8192, 131072) if args.workload == "rect131072" else (n, n, n)
    if args.dims:
        M, L, N = (int(x) for x in args.dims.split(","))
        assert args.workload == "mul" and min(M, L, N) > 0
    else:
        assert n % 64 == 0
    wl, w = (L + 63) // 64, (N + 63) // 64  # words per row of A, and of B / C
    seeds = (5, 6) if args.workload == "rect131072" else (3, 4)
    if args.seeds:
        seeds = tuple(int(x) for x in args.seeds.split(","))
    auto = args.variant == "auto"
    if auto:
        args.variant = sharding.default_variant(world, M, L, N)
    plan = None
    if args.variant == "strassen" and multi:
        plan = m4ri_amd.shard_plan(world, M, L, N, args.shard_levels)
        if (plan.M, plan.L, plan.N) != (M, L, N) and not peer:  # the slab-cyclic layout of this script holds unpadded slabs only (the library pads its own)
            if not auto:
                raise SystemExit(f"--variant strassen: {M}x{L}x{N} needs padding to {plan.M}x{plan.L}x{plan.N} in the slab-cyclic layout; "
                                 "bench.py runs such sizes as row slabs (mzd_mul_mp pads them itself)")
            args.variant = "slabs"

    def words(count):
        return torch.empty(max(1, int(count)), dtype=torch.int64, device="cuda")

    def fence():
        torch.cuda.synchronize()
        if peer:
            m4ri_amd.multi_sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    exchange = None
    if dist is not None:
        exchange = sharding.torch_exchange(dist, staged_device=("cuda" if args.backend == "gloo" else None))
    A = B = Cfull = None
    config_extra = {}
    phase_steps = None            # per buffer slot: an object with start() / multiply() / finish() (sharding.run_products)
    last = {"slot": 0}            # the slot that holds the C of the last product
    inflight = args.inflight if (multi and not peer and args.layout == "distributed" and args.variant in ("slabs", "strassen")) else 1
    if args.variant == "blocks" and args.layout == "distributed" and multi:
        args.layout = "owner"  # the blocks variant scatters from rank 0 by construction
    need_full_inputs = not multi or ((args.layout == "owner" or args.variant == "blocks") and not peer)
    if need_full_inputs and (not multi or rank == 0):
        A = torch.empty((M, wl), dtype=torch.int64, device="cuda")
        B = torch.empty((L, w), dtype=torch.int64, device="cuda")
        m4ri_amd.fill_dev(A.data_ptr(), wl, M, L, seeds[0], stream)
        m4ri_amd.fill_dev(B.data_ptr(), w, L, N, seeds[1], stream)

    # ---------------------------------------------------------------- N == 1 -----------------------
    if not multi:
        Cfull = torch.empty((M, w), dtype=torch.int64, device="cuda")

        def step():
            if args.workload == "leaf16384":
                m4ri_amd.m4rm_dev(Cfull.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, M, L, N, False, 0, stream)
            else:
                m4ri_amd.mul_dev(Cfull.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, M, L, N, False, args.cutoff, stream)
        per_rank_product = [M, L, N]
        config_extra["parallelism"] = "1 GPU"

    # ---------------------------------------------------------------- N > 1, one process, all GPUs: the schedules behind the C boundary
    elif peer:
        if args.variant not in ("slabs", "strassen"):
            raise SystemExit("--transport peer runs the slabs and the strassen schedule (m4ri_amd_dmat_mul)")
        v = m4ri_amd.VARIANT_SLABS if args.variant == "slabs" else m4ri_amd.VARIANT_STRASSEN
        if v == m4ri_amd.VARIANT_STRASSEN:
            lay = m4ri_amd.LAYOUT_CYCLIC2 if plan.levels == 2 else m4ri_amd.LAYOUT_CYCLIC1
        else:
            lay = m4ri_amd.LAYOUT_ROWS
        dA, dB, dC = m4ri_amd.Dmat(M, L, lay).fill(seeds[0]), m4ri_amd.Dmat(L, N, lay).fill(seeds[1]), m4ri_amd.Dmat(M, N, lay)
        issue_s = [0.0]

        def step():   # asynchronous: returns when every rank's host thread has issued its part (the devices work on)
            ta = time.perf_counter()
            m4ri_amd.dmat_mul(dC, dA, dB, False, args.cutoff, v)
            issue_s[0] += time.perf_counter() - ta
        if v == m4ri_amd.VARIANT_STRASSEN:
            per_rank_product = [plan.bm, plan.bl, plan.cwn * 64]
        else:
            per_rank_product = [sharding.slab_rows(M, world), L, N]
        config_extra.update({"parallelism": (f"strassen-sharded x{world}" if v == m4ri_amd.VARIANT_STRASSEN else f"row slabs x{world} + gather of B")
                                            + ", one process, peer copies", "variant": args.variant, "layout": "distributed",
                             "devices": m4ri_amd.get_devices(), "virtual_ranks_sharing_devices": len(set(m4ri_amd.get_devices())) < world,
                             "collective": "hipMemcpyPeerAsync pulls on per-device copy streams, HIP events between ranks; one host thread per rank (multi.hip)"})

    # ---------------------------------------------------------------- N > 1, row slabs + all-gather of B
    elif args.variant == "slabs":
        # slabs of ceil(rows / world) rows; the last one(s) may be short or empty (sharding.slab_cuts).  The gathered B is
        # padded to world * kb rows so that the ONE collective keeps equal pieces; the product only reads its first L rows
        rc, bc = sharding.slab_cuts(M, world), sharding.slab_cuts(L, world)
        ka, kb = sharding.slab_rows(M, world), sharding.slab_rows(L, world)
        mr, lr = rc[rank + 1] - rc[rank], bc[rank + 1] - bc[rank]   # rows of this rank's slab of A / C, and of B
        staged = args.backend == "gloo"
        Bfull_slots = [torch.empty((world * kb, w), dtype=torch.int64, device="cuda") for _ in range(inflight)]
        Cs_slots = [torch.empty((max(mr, 1), w), dtype=torch.int64, device="cuda")[:mr] for _ in range(inflight)]
        Bfull, Cs = Bfull_slots[0], Cs_slots[0]
        As = torch.empty((max(mr, 1), wl), dtype=torch.int64, device="cuda")[:mr]
        if args.layout == "distributed":           # the slabs are where the inputs live
            Bs = torch.zeros((kb, w), dtype=torch.int64, device="cuda")
            if mr:
                m4ri_amd.fill_rows_dev(As.data_ptr(), wl, rc[rank], mr, L, seeds[0], stream)
            if lr:
                m4ri_amd.fill_rows_dev(Bs.data_ptr(), w, bc[rank], lr, N, seeds[1], stream)
        else:
            Bs = None
            if rank == 0:
                Cfull = torch.empty((M, w), dtype=torch.int64, device="cuda")
        # the all-gather of B under the first product: needs every slab boundary of B on a word of A's rows, and pays when a rank's
        # own slab is a large part of the inner dimension (few ranks); --slab-overlap 0/1 overrides
        aligned = all(c % 64 == 0 for c in bc[:-1]) and lr > 0
        slab_overlap = args.layout == "distributed" and aligned and (world <= 4 if args.slab_overlap < 0 else bool(args.slab_overlap))

        def step():   # one product at a time, always on slot 0
            Cs, Bfull = Cs_slots[0], Bfull_slots[0]
            if args.layout == "owner":             # rank 0 scatters the slabs of A and broadcasts B; C is gathered
                sends, recvs = [], []
                for r in range(1, world):
                    if rank == 0:
                        sends += [(r, A[rc[r]:rc[r + 1]].reshape(-1)), (r, B.reshape(-1))]
                    elif rank == r:
                        recvs += [(0, As.reshape(-1)), (0, Bfull[:L].reshape(-1))]
                if rank == 0:
                    As.copy_(A[:mr])
                    Bfull[:L].copy_(B)
                exchange([x for x in sends if x[1].numel()], [x for x in recvs if x[1].numel()])
            elif slab_overlap:
                # the ONE collective of the variant runs under the product with the rank's own slab of B: C_r = A_r[:, own] * B_own
                # first (nothing to wait for), then += A_r[:, before] * B[before] and A_r[:, after] * B[after] from the gathered B
                pending = sharding.all_gather_rows(dist, Bfull, Bs, staged=staged, async_op=True)
                first = True
                for k0, k1, own in sharding.slab_product_pieces(bc, rank):
                    if not own and pending is not None:
                        pending.wait()
                        pending = None
                    if mr:
                        bsrc = Bs.data_ptr() if own else Bfull.data_ptr() + 8 * k0 * w
                        m4ri_amd.mul_dev(Cs.data_ptr(), w, As.data_ptr() + 8 * (k0 // 64), wl, bsrc, w, mr, k1 - k0, N, not first, args.cutoff, stream)
                        first = False
                if pending is not None:
                    pending.wait()
            else:
                sharding.all_gather_rows(dist, Bfull, Bs, staged=staged)   # the ONE collective of the variant
                if mr:
                    m4ri_amd.mul_dev(Cs.data_ptr(), w, As.data_ptr(), wl, Bfull.data_ptr(), w, mr, L, N, False, args.cutoff, stream)
            if mr and args.layout == "owner":
                m4ri_amd.mul_dev(Cs.data_ptr(), w, As.data_ptr(), wl, Bfull.data_ptr(), w, mr, L, N, False, args.cutoff, stream)
            if args.layout == "owner":
                sends, recvs = [], []
                for r in range(1, world):
                    if rc[r + 1] == rc[r]:
                        continue
                    if rank == 0:
                        recvs.append((r, Cfull[rc[r]:rc[r + 1]].reshape(-1)))
                    elif rank == r:
                        sends.append((0, Cs.reshape(-1)))
                if rank == 0:
                    Cfull[:mr].copy_(Cs)
                exchange(sends, recvs)
        class SlabStep:   # the throughput form: the whole product against the gathered B, the all-gather of the NEXT product's B under it
            def __init__(self, slot):
                self.slot, self.pending = slot, None

            def start(self):
                self.pending = sharding.all_gather_rows(dist, Bfull_slots[self.slot], Bs, staged=staged, async_op=True)

            def multiply(self):
                self.pending.wait()
                if mr:
                    m4ri_amd.mul_dev(Cs_slots[self.slot].data_ptr(), w, As.data_ptr(), wl, Bfull_slots[self.slot].data_ptr(), w, mr, L, N, False,
                                     args.cutoff, stream)

            def finish(self):
                pass
        if inflight > 1:
            phase_steps = [SlabStep(slot) for slot in range(inflight)]
        per_rank_product = [ka, L, N]
        config_extra.update({"parallelism": f"row slabs x{world} + all-gather of B", "variant": "slabs", "layout": args.layout,
                             "slab_rows": [ka, kb], "all_gather_under_first_product": bool(slab_overlap),
                             "bytes_over_links_per_step": 8 * kb * w * (world - 1) * (1 if args.layout == "distributed" else 0),
                             "collective": "all_gather_into_tensor(B)" if args.layout == "distributed" else "batched send/recv scatter + gather"})

    # ---------------------------------------------------------------- N > 1, Strassen sub-products --
    elif args.variant == "strassen":
        names = {"local_a": m4ri_amd.BUF_LOCAL_A, "local_b": m4ri_amd.BUF_LOCAL_B, "local_c": m4ri_amd.BUF_LOCAL_C,
                 "child_a": m4ri_amd.BUF_CHILD_A, "child_b": m4ri_amd.BUF_CHILD_B, "slabs_p": m4ri_amd.BUF_SLABS_P,
                 "oper_a": m4ri_amd.BUF_OPER_A, "oper_b": m4ri_amd.BUF_OPER_B, "prod": m4ri_amd.BUF_PROD}
        bufs = {k: words(m4ri_amd.shard_buffer_words(plan, rank, wh)) for k, wh in names.items()}
        # a second product in flight has its own children / operands / products / slabs / result; the inputs are shared
        slot_bufs = [bufs] + [{k: (bufs[k] if k in ("local_a", "local_b") else words(m4ri_amd.shard_buffer_words(plan, rank, wh)))
                               for k, wh in names.items()} for _ in range(inflight - 1)]
        runs_a, runs_b = sharding.local_rows(plan, rank, 0), sharding.local_rows(plan, rank, 1)
        sa, sb = runs_a[0][1], runs_b[0][1]
        # units per sub-product whose transport runs under the products (sharding.run_strassen_sharded).  Default: two ROW chunks when
        # one product per rank is all there is to hide transfers behind and its halves keep the engine's Strassen depth (measured:
        # two 16384 x 32768 x 32768 halves cost 1.00 - 1.03 of the whole).  Column chunks on top (--overlap 2x2) expose less on the
        # links but their quarter-size leaf batches fill the chip 1.5 times: 4 x 1.29 ms against 4.10 ms, a net loss
        # (profiles/r03_rank_shapes_timing.log) -- available, not the default
        if args.overlap:
            chunks = sharding.parse_chunks(args.overlap)
        else:
            chunks = (2, 1) if (plan.levels == 1 and plan.bm >= 4 * sharding.ENGINE_MIN_HALF[0]) else (1, 1)
        if args.layout == "distributed":  # the slabs are where the inputs live: fill them straight from the streams
            for b, (g0, rows) in enumerate(runs_a):
                m4ri_amd.fill_rows_dev(bufs["local_a"].data_ptr() + 8 * b * sa * wl, wl, g0, rows, L, seeds[0], stream)
            for b, (g0, rows) in enumerate(runs_b):
                m4ri_amd.fill_rows_dev(bufs["local_b"].data_ptr() + 8 * b * sb * w, w, g0, rows, N, seeds[1], stream)
        if args.layout == "owner" and rank == 0:
            Cfull = torch.empty((M, w), dtype=torch.int64, device="cuda")

        def scatter_from_owner():
            sends, recvs = [], []
            for r in range(world):
                for key, full, runs, width in (("local_a", A, sharding.local_rows(plan, r, 0), wl), ("local_b", B, sharding.local_rows(plan, r, 1), w)):
                    s = runs[0][1]
                    for b, (g0, rows) in enumerate(runs):
                        if rows == 0:
                            continue
                        if rank == 0 and r == 0:
                            bufs[key][b * s * width:(b * s + rows) * width].copy_(full[g0:g0 + rows].reshape(-1))
                        elif rank == 0:
                            sends.append((r, full[g0:g0 + rows].reshape(-1)))
                        elif rank == r:
                            recvs.append((0, bufs[key][b * s * width:(b * s + rows) * width]))
            exchange(sends, recvs)

        def gather_to_owner():
            sends, recvs = [], []
            for r in range(world):
                runs = sharding.local_rows(plan, r, 0)
                s = runs[0][1]
                for b, (g0, rows) in enumerate(runs):
                    if rows == 0:
                        continue
                    if rank == 0 and r == 0:
                        Cfull[g0:g0 + rows].reshape(-1).copy_(bufs["local_c"][b * s * w:(b * s + rows) * w])
                    elif rank == 0:
                        recvs.append((r, Cfull[g0:g0 + rows].reshape(-1)))
                    elif rank == r:
                        sends.append((0, bufs["local_c"][b * s * w:(b * s + rows) * w]))
            exchange(sends, recvs)

        def sharded_step(sb_, chunks_):   # the three phases of one product on one slot's buffers
            def do_down():
                m4ri_amd.shard_down_dev(plan, rank, sb_["local_a"].data_ptr(), wl, sb_["local_b"].data_ptr(), w,
                                        sb_["child_a"].data_ptr(), sb_["child_b"].data_ptr(), stream)

            def do_product(jl, j, row0=0, rows=None, w0=0, w1=None):
                rows = plan.bm if rows is None else rows
                w1 = plan.cwn if w1 is None else w1
                m4ri_amd.mul_dev(sb_["prod"].data_ptr() + 8 * ((jl * plan.bm + row0) * plan.cwn + w0), plan.cwn,
                                 sb_["oper_a"].data_ptr() + 8 * (jl * plan.bm + row0) * plan.cwl, plan.cwl,
                                 sb_["oper_b"].data_ptr() + 8 * (jl * plan.bl * plan.cwn + w0), plan.cwn,
                                 rows, plan.bl, (w1 - w0) * 64, False, args.cutoff, stream)

            def do_up():
                m4ri_amd.shard_up_dev(plan, rank, sb_["slabs_p"].data_ptr(), sb_["local_c"].data_ptr(), w, False, stream)
            return sharding.StrassenShardedStep(plan, rank, sb_, do_down, do_product, do_up, exchange, lambda d, s: d.copy_(s), chunks=chunks_)
        # one product at a time (step(), `latency_ms`): row chunks hide part of its own transport.  Two products in flight: the neighbours'
        # multiplications hide all of it, so the sub-products stay whole (their halves cost up to 3 % more than the whole)
        single_step = sharded_step(slot_bufs[0], chunks)
        chunks_loop = chunks if (args.overlap or inflight == 1) else (1, 1)
        if inflight > 1:
            phase_steps = [sharded_step(sb_, chunks_loop) for sb_ in slot_bufs]

        def step():
            if args.layout == "owner":
                scatter_from_owner()
            single_step.start()
            single_step.multiply()
            single_step.finish()
            if args.layout == "owner":
                gather_to_owner()
        per_rank_product = [plan.bm, plan.bl, plan.cwn * 64]
        moved = sum(pc.words * 8 for side, j, r, pc in sharding.strassen_pieces(plan, (0, 1, 2)) if pc.holder != pc.owner)
        config_extra.update({"parallelism": f"strassen-sharded x{world}", "variant": "strassen", "layout": args.layout, "sharded_levels": plan.levels,
                             "sub_products": plan.nprod, "sub_products_on_busiest_rank": len(sharding.owned_products(plan, 0)),
                             "bytes_over_links_per_step": moved, "links_used": world * (world - 1), "overlap_chunks": list(chunks),
                             "overlap_chunks_in_the_timed_loop": list(chunks_loop),
                             "collective": f"batched isend/irecv (one group per batch: operands out per round and row / column chunk, products back "
                                           f"per unit; {len(sharding.chunk_bounds(plan, chunks[0]))} x {len(sharding.column_bounds(plan, chunks[1]))} unit(s) "
                                           f"x {-(-plan.nprod // world)} round(s))",
                             "scatter_gather_bytes_per_step": (8 * (M * wl + L * w + M * w) * (world - 1) // world) if args.layout == "owner" else 0})

    # ---------------------------------------------------------------- N > 1, blocks of C -----------
    else:
        grid = tuple(int(x) for x in args.grid.split(",")) if args.grid else ((world, 1, 1) if args.workload == "rect131072" else None)
        bplan = sharding.make_plan(world, rank, M, L, N, grid=grid)
        r0, r1 = bplan.row_range()
        c0, c1 = bplan.col_range()
        k0, k1 = bplan.inner_range()
        Ablk = torch.empty((r1 - r0, (k1 - k0) // 64), dtype=torch.int64, device="cuda")
        Bblk = torch.empty((k1 - k0, (c1 - c0) // 64), dtype=torch.int64, device="cuda")
        P = torch.empty((r1 - r0, (c1 - c0) // 64), dtype=torch.int64, device="cuda")
        pw = P.shape[1]
        gh = bplan.grid[2]
        cuts = sharding.ShardPlan._cuts(r1 - r0, gh, 1)
        recv_buf = torch.empty((cuts[bplan.h + 1] - cuts[bplan.h], pw), dtype=torch.int64, device="cuda") if gh > 1 else None
        if rank == 0:
            Cfull = torch.empty((M, w), dtype=torch.int64, device="cuda")

        def plan_of(r):
            return sharding.make_plan(world, r, M, L, N, grid=grid)

        def scatter_blocks():  # mp.c:191-204's zero-copy windows become real transfers: A_ih, B_hj to rank (i, j, h)
            sends, recvs, keep = [], [], []
            for r in range(world):
                pr = plan_of(r)
                (a0, a1), (b0, b1), (h0, h1) = pr.row_range(), pr.col_range(), pr.inner_range()
                if rank == 0:
                    ab = A[a0:a1, h0 // 64:h1 // 64]
                    bb = B[h0:h1, b0 // 64:b1 // 64]
                    if r == 0:
                        Ablk.copy_(ab)
                        Bblk.copy_(bb)
                    else:
                        ab, bb = ab.contiguous(), bb.contiguous()
                        keep += [ab, bb]
                        sends += [(r, ab.reshape(-1)), (r, bb.reshape(-1))]
                elif rank == r:
                    recvs += [(0, Ablk.reshape(-1)), (0, Bblk.reshape(-1))]
            exchange(sends, recvs)

        def multiply(_r0, _r1, _k0, _k1, _c0, _c1):
            m4ri_amd.mul_dev(P.data_ptr(), pw, Ablk.data_ptr(), Ablk.shape[1], Bblk.data_ptr(), pw, r1 - r0, k1 - k0, c1 - c0, False, args.cutoff, stream)

        def send_recv(partner, send_rows, recv_rows):
            exchange([(partner, P[send_rows[0]:send_rows[1]].reshape(-1))], [(partner, recv_buf.reshape(-1))])
            return recv_buf

        def xor_rows(rows, got):
            ptr = P.data_ptr() + 8 * rows[0] * pw
            m4ri_amd.xor_dev(ptr, pw, ptr, pw, got.data_ptr(), pw, rows[1] - rows[0], (c1 - c0), stream)

        def gather_blocks(region):
            sends, recvs, tmp = [], [], []
            for r in range(world):
                pr = plan_of(r)
                o0, o1 = pr.owned_rows_after_reduce()
                b0, b1 = pr.col_range()
                if rank == 0 and r == 0:
                    Cfull[o0:o1, b0 // 64:b1 // 64].copy_(P[o0 - r0:o1 - r0])
                elif rank == 0:
                    t = torch.empty((o1 - o0, (b1 - b0) // 64), dtype=torch.int64, device="cuda")
                    tmp.append((t, o0, o1, b0, b1))
                    recvs.append((r, t.reshape(-1)))
                elif rank == r:
                    sends.append((0, P[region[0] - r0:region[1] - r0].reshape(-1)))
            exchange(sends, recvs)
            for t, o0, o1, b0, b1 in tmp:
                Cfull[o0:o1, b0 // 64:b1 // 64].copy_(t)

        def step():
            scatter_blocks()
            region = sharding.run_sharded(bplan, multiply, xor_rows, send_recv)
            gather_blocks(region)
        per_rank_product = [r1 - r0, k1 - k0, c1 - c0]
        config_extra.update({"parallelism": f"blocks {list(bplan.grid)}", "variant": "blocks", "layout": "owner", "grid": list(bplan.grid),
                             "collective": "batched isend/irecv: scatter from rank 0, pairwise XOR exchange, gather to rank 0"})
    if dist is not None:
        config_extra.update({"backend": "nccl (RCCL)" if args.backend == "nccl" else "gloo (pieces staged through the host: development aid)",
                             "ranks": dist.get_world_size()})

    # ---------------------------------------------------------------- probe mode (under rocprofv3) --
    if args.probe:
        step()
        torch.cuda.synchronize()
        step()
        torch.cuda.synchronize()
        return

    # ---------------------------------------------------------------- timing ------------------------
    def run_steps(n, marks=None, pipelined=False):
        """n products back to back: one at a time through step(), or -- pipelined -- two in flight over two buffer slots."""
        before = (lambda k: marks[k].record()) if marks is not None else None
        if not pipelined:
            for k in range(n):
                if before is not None:
                    before(k)
                step()
            last["slot"] = 0
        else:
            sharding.run_products(lambda k: phase_steps[k % inflight], n, inflight, before)
            last["slot"] = (n - 1) % inflight

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cuda")
        if args.backend == "gloo":
            tt = tt.cpu()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    run_steps(args.warmup)
    fence()
    m4ri_amd.set_profiling(2)  # leaf launches bracketed by HIP events on their stream, accumulated over all steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if peer:
        issue_s[0] = 0.0
    t0 = time.perf_counter()
    run_steps(args.steps, marks)
    marks[args.steps].record()
    t_posted = time.perf_counter()   # every step is posted; the devices may still be working
    fence()
    t1 = time.perf_counter()
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)) if not peer else None
    stats = m4ri_amd.get_stats()  # last product's schedule + the leaf launch durations of ALL timed steps
    m4ri_amd.set_profiling(0)
    elapsed = max_over_ranks(t1 - t0)
    # how long the HOST needs to post one step, no fence inside: must stay well below the step for the devices never to wait for it
    host_issue_ms = 1e3 * max_over_ranks((issue_s[0] if peer else (t_posted - t0)) / args.steps)
    ms_per_step = 1e3 * elapsed / args.steps
    ops = float(M) * L * N  # classical bit multiply-accumulates of the WHOLE product (AND+XOR = 1 op)
    value = ops * args.steps / elapsed
    pipelined = None
    if phase_steps is not None and inflight > 1:
        # a second, separately named measurement: the throughput of a STREAM of independent products, two in flight (the transport of
        # the neighbouring products under the multiplications of the current one).  Never the headline: the metric is one mzd_mul
        run_steps(2, None, True)
        fence()
        tp0 = time.perf_counter()
        run_steps(args.steps, None, True)
        fence()
        tp = max_over_ranks(time.perf_counter() - tp0)
        pipelined = {"pipelined_value": ops * args.steps / tp, "pipelined_ms_per_step": 1e3 * tp / args.steps}
    if multi and not peer and args.variant == "slabs":
        Cs, Bfull = Cs_slots[last["slot"]], Bfull_slots[last["slot"]]
    if multi and not peer and args.variant == "strassen":
        bufs = slot_bufs[last["slot"]]

    # ---------------------------------------------------------------- correctness of what was timed --
    verified = None
    if not multi and not args.no_verify and args.workload != "leaf16384":
        want = golden_sha("mul", M, L, N, seeds)
        if want is not None:
            got = sha_of_device_rows(Cfull)
            verified = {"sha256": got, "matches_reference": got == want,
                        "what": "C of the last timed step vs the real reference's product of the same inputs (tests/golden/sha256.json)"}
            if got != want:
                print(json.dumps({"error": "C differs from the reference", "sha256": got, "expected": want}), flush=True)
                raise SystemExit(4)
    Chost = None
    if peer and not args.no_verify:
        want = golden_sha("mul", M, L, N, seeds)
        if want is not None:
            Chost = dC.download()   # every device its own rows, outside the timed region
            got = hashlib.sha256(Chost.masked().tobytes()).hexdigest()
            verified = {"sha256": got, "matches_reference": got == want,
                        "what": f"C of the last timed step, downloaded from the {world} ranks, vs the real reference's product of the same inputs "
                                "(tests/golden/sha256.json)"}
            if got != want:
                print(json.dumps({"error": "C differs from the reference", "sha256": got, "expected": want}), flush=True)
                raise SystemExit(4)
    if multi and not peer and not args.no_verify and args.workload != "leaf16384":
        # the distributed result against the reference's: gather the ranks' rows of C on rank 0 (outside the timed region,
        # over the same transport) and compare its SHA-256 with the golden one, when tests/golden holds one for this product
        want = golden_sha("mul", M, L, N, seeds)
        if want is not None:
            if Cfull is None and rank == 0:
                Cfull = torch.empty((M, w), dtype=torch.int64, device="cuda")
            if args.layout == "distributed" and args.variant in ("slabs", "strassen"):
                def rows_of(r):
                    if args.variant == "slabs":
                        return [(rc[r], rc[r + 1] - rc[r], 0)]
                    rr = sharding.local_rows(plan, r, 0)
                    return [(g0, rows, b * rr[0][1] * w) for b, (g0, rows) in enumerate(rr)]
                mine = Cs.reshape(-1) if args.variant == "slabs" else bufs["local_c"]
                sends, recvs = [], []
                for r in range(world):
                    for g0, rows, off in rows_of(r):
                        if rows == 0:
                            continue
                        if rank == 0 and r == 0:
                            Cfull[g0:g0 + rows].reshape(-1).copy_(mine[off:off + rows * w])
                        elif rank == 0:
                            recvs.append((r, Cfull[g0:g0 + rows].reshape(-1)))
                        elif rank == r:
                            sends.append((0, mine[off:off + rows * w]))
                exchange(sends, recvs)
            if rank == 0:
                torch.cuda.synchronize()
                got = sha_of_device_rows(Cfull)
                verified = {"sha256": got, "matches_reference": got == want,
                            "what": f"C of the last timed step, gathered from the {world} ranks, vs the real reference's product of the same inputs "
                                    "(tests/golden/sha256.json)"}
                if got != want:
                    print(json.dumps({"error": "C differs from the reference", "sha256": got, "expected": want}), flush=True)
                    raise SystemExit(4)
    if args.check and multi:
        fullA = torch.empty((M, wl), dtype=torch.int64, device="cuda")
        fullB = torch.empty((L, w), dtype=torch.int64, device="cuda")
        m4ri_amd.fill_dev(fullA.data_ptr(), wl, M, L, seeds[0], stream)
        m4ri_amd.fill_dev(fullB.data_ptr(), w, L, N, seeds[1], stream)
        full = torch.empty((M, w), dtype=torch.int64, device="cuda")
        m4ri_amd.mul_dev(full.data_ptr(), w, fullA.data_ptr(), wl, fullB.data_ptr(), w, M, L, N, False, 0, stream)
        torch.cuda.synchronize()
        if peer:
            Chost = Chost if Chost is not None else dC.download()
            ok = bool(np.array_equal(Chost.valid_words().view(np.int64), full.cpu().numpy()))
            what = "the whole C, downloaded"
        elif args.variant == "slabs":
            ok = bool(torch.equal(Cs, full[rc[rank]:rc[rank + 1]]))
            what = f"rows {rc[rank]}:{rc[rank + 1]} of C"
        elif args.variant == "strassen":
            ok = True
            for b, (g0, rows) in enumerate(runs_a):
                ok = ok and bool(torch.equal(bufs["local_c"][b * sa * w:(b * sa + rows) * w].reshape(rows, w), full[g0:g0 + rows]))
            what = f"slabs {[(g0, g0 + rows) for g0, rows in runs_a]} of C"
        else:
            o0, o1 = bplan.owned_rows_after_reduce()
            ok = bool(torch.equal(P[o0 - r0:o1 - r0], full[o0:o1, c0 // 64:c1 // 64]))
            what = f"rows {o0}:{o1} cols {c0}:{c1}"
        if rank == 0 and Cfull is not None and not peer:
            ok = ok and bool(torch.equal(Cfull, full))
            what += " + the gathered C"
        print(f"[check] rank {rank} {config_extra.get('parallelism')} {what} -> {'OK' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            raise SystemExit(3)

    if rank == 0:
        launches = max(1, int(stats.cum_leaf_launches))
        leaf_launch_ms = stats.cum_leaf_ms / launches               # mean over every leaf launch of the timed steps
        leaf_launch_bytes = stats.leaf_bytes / max(1, stats.leaf_launches)
        achieved = leaf_launch_bytes / (leaf_launch_ms * 1e-3) / 1e9 if leaf_launch_ms > 0 else 0.0
        leaf_ops = float(stats.leaf_m) * stats.leaf_l * stats.leaf_n * stats.leaf_products
        leaf_ms_per_product = leaf_launch_ms * max(1, int(stats.leaf_launches))
        traffic, traffic_detail = None, "not measured for this configuration"
        if not multi and args.workload == "mul" and not args.no_traffic:
            traffic, traffic_detail = measure_leaf_traffic(n, args.cutoff)
        out = {
            "metric": "gf2_matmul_n3_equiv_bitops_per_sec",
            "value": value,
            "unit": "bit-op/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": (f"mzd_mul {M}x{L}x{N} (not a BASELINE.json configuration): Strassen-Winograd over M4RM leaves"
                             if args.workload == "mul" and (M, L, N) != (65536, 65536, 65536) else
                             f"mzd_mul {n}x{n}x{n} (BASELINE.json configs[2]/[3]): Strassen-Winograd over M4RM leaves"
                             if args.workload == "mul" else
                             f"mzd_mul {M}x{L}x{N} (BASELINE.json configs[4])" if args.workload == "rect131072"
                             else f"mzd_mul_m4rm {n}^3 leaf only (BASELINE.json configs[1])"),
                "m": M, "l": L, "n": N,
                "ops_counted": "m*l*n bit multiply-accumulates (one AND+XOR = 1 op), classical count credited to Strassen",
                "input": f"splitmix64 seeds {seeds[0]} (A), {seeds[1]} (B), uniform bits, resident in HBM"
                         + ("" if not multi else " (slab-cyclic over the ranks)" if args.layout == "distributed" else " of rank 0; C gathered to rank 0"),
                "per_rank_product": per_rank_product,
                "strassen_levels": int(stats.levels),
                # the engine's own plan for the per-rank product: rows in blocks [rows, levels], largest first (one block = one product)
                "row_blocks": ([list(b) for b in m4ri_amd.plan_row_blocks(*per_rank_product)] if args.workload != "leaf16384" and not args.cutoff else None),
                "leaf_shape": [int(stats.leaf_m), int(stats.leaf_l), int(stats.leaf_n)],
                "leaf_products_per_rank": int(stats.leaf_products),
                "workspace_GiB": stats.workspace_bytes / 2 ** 30,
                **config_extra,
            },
            "roofline": {
                # the resource that binds the kernel is the LDS array (`lds` below); achieved / peak / frac are the HBM figures SURVEY.md 8(d) asks for
                "bound": "lds",
                "kernel": LEAF_KERNELS.get(int(stats.leaf_gen), "?") + " (the M4RM leaf; HIP events around every launch on its stream, mean over all timed steps)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "hbm_frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "launch_ms": leaf_launch_ms,
                "launches_timed": int(stats.cum_leaf_launches),
                "launches_per_product": int(stats.leaf_launches),
                "algorithmic_bytes_per_launch": leaf_launch_bytes,
                "leaf_bitops_per_sec": (leaf_ops / (leaf_ms_per_product * 1e-3)) if leaf_ms_per_product > 0 else 0.0,
                "aux_pass_bytes_per_product": stats.aux_bytes,
                "lds": lds_model(int(stats.leaf_gen), int(stats.leaf_m), int(stats.leaf_l), int(stats.leaf_n),
                                 int(stats.leaf_products) // max(1, int(stats.leaf_launches)), leaf_launch_ms),
                "note": "the leaf is LDS-bound by design (table gathers at 256 B/clk/CU), not HBM-bound: "
                        "its algorithmic HBM bytes are ~1e-3 of its LDS traffic, so frac is small; the `lds` object "
                        "prices the launch against the LDS-array cycles it needs (DESIGN.md 3.1)",
            },
        }
        lds = out["roofline"]["lds"]
        if lds and isinstance(traffic_detail, dict) and traffic_detail.get("gui_active_cycles"):
            # in cycles the clock drops out: LDS-array cycles the launch needs / cycles it took (same launch, profiled pass)
            need = lds["bound_ms"] * 1e-3 * PEAK_CLOCK_HZ
            lds["frac_in_cycles"] = need / traffic_detail["gui_active_cycles"]
            lds["effective_clock_hz"] = traffic_detail["effective_clock_hz"]
            lds["note"] = ("frac prices the launch at the 2.4 GHz peak clock, frac_in_cycles at the clock it really ran at (GRBM_GUI_ACTIVE / duration): "
                           "the kernel sits on the chip's power limit (measured: 1338 W of a 1400 W socket cap, the firmware's package-power limiter active 82 % of the launch's time, "
                           "no thermal limiter -- profiles/r04_leaf_power/).  Of the cycles with the LDS idle ~10 % are bubbles of the gather pipeline itself "
                           "(gathers alone: 90 % busy) and ~4.5 % the stage barrier (profiles/r03_leaf_decomposition/README.md)")
        if step_ms:
            out["step_ms_min"], out["step_ms_median"] = step_ms[0], step_ms[len(step_ms) // 2]
        out["host_issue_ms_per_step"] = host_issue_ms
        if multi:
            out["config"]["inflight"] = 1
            out["config"]["transport"] = "peer" if peer else "rccl"
            if pipelined is not None:
                out.update(pipelined)
                out["pipelined_note"] = ("a stream of independent products, two in flight (transport of the neighbouring products under the multiplications): "
                                         "throughput of that stream; `value` / `ms_per_step` are one product at a time")
            if peer:
                ms = m4ri_amd.multi_stats()
                out["config"].update({"schedule_stats": {"variant": m4ri_amd.VARIANT_NAMES.get(ms.variant), "sharded_levels": ms.levels, "sub_products": ms.sub_products,
                                                         "row_chunks": ms.chunks, "gather_under_first_product": bool(ms.overlap),
                                                         "operands_converted": ms.converted, "bytes_over_links_per_step": ms.link_bytes},
                                      "timeline_ms_last_step": {str(r): m4ri_amd.multi_timeline(r) for r in range(world)},
                                      "timeline_marks": ("strassen: down pass done; per row chunk: operands in, product done; result slabs in; up pass done"
                                                         if ms.variant == m4ri_amd.VARIANT_STRASSEN else "slabs: gather done; first product done; all done")})
        if verified is not None:
            out["verified"] = verified
        if args.workload != "leaf16384":
            copy_gbs = measured_copy_gbs()
            # the whole product against the HBM roofline in SURVEY.md 8(d)'s terms (schedule bytes of
            # this rank's products / step time); the compulsory bytes beside it
            pm, pl, pn = per_rank_product
            nprod_rank = 1 if (not multi or args.variant != "strassen") else len(sharding.owned_products(plan, 0))
            if peer and args.variant == "strassen":   # the library's own padding (every dimension to 256 bits) and row chunks
                pm, pl, pn = plan.bm, plan.bl, plan.cwn * 64
            bs = float(bytes_sched(pm, pl, pn, int(stats.levels))) * nprod_rank
            out["roofline_schedule"] = {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "bytes_sched_per_rank": bs, "levels": int(stats.levels),
                "bytes_compulsory_per_rank": float(bytes_sched(pm, pl, pn, 0)) * nprod_rank,
                "achieved": bs / (ms_per_step * 1e-3) / 1e9,
                "frac": bs / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "bytes_moved_by_our_fused_passes": stats.aux_bytes + stats.leaf_bytes,
                "copy_peak": copy_gbs, "frac_of_copy_peak": bs / (ms_per_step * 1e-3) / 1e9 / copy_gbs if copy_gbs else None,
                "note": "unfused reference schedule bytes (15 quadrant adds/level) of the rank's sub-product(s) over the measured step time; "
                        "our fused multi-level passes move far fewer bytes; copy_peak = this GPU's measured "
                        "device-to-device copy rate (read + write bytes)",
            }
        if not multi and args.workload != "leaf16384" and not args.no_api:
            try:
                out["api"] = host_api_timing(A, B, M, L, N, args.cutoff)
                out["api_ms"] = out["api"]["c_given_ms_min"]
            except Exception as e:  # reported beside the number, never required for it
                out["api"] = {"error": repr(e)}
        if not multi and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "bit-op/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
