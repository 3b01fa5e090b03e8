#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X:

    GF(2) n^3-equivalent bit-ops/s and wall-clock of one n x n x n mzd_mul, n = 65536,
    at 1/2/4/8 GPUs (strong scaling: the total work is fixed).

A "step" is one whole product C = A*B on device-resident, synthetic (splitmix64, density 1/2)
operands: Strassen-Winograd levels over batched M4RM leaves, everything through libm4ri_amd.so's
C ABI.  Inputs are in HBM before the timed region starts; C stays in HBM (distributed over the ranks
that own its blocks when N > 1).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 65536] [--workload mul|leaf16384]

For N > 1 the driver launches one process per GPU with torch.distributed.run (RCCL); the product is
decomposed by m4ri_amd/sharding.py (block products, one pairwise XOR exchange at N = 8).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline"     the dominant kernel (the M4RM leaf launch) against the HBM roofline, duration
                 measured with HIP events on the launch stream inside the timed region;
  "cpu_baseline" the real reference M4RI (oracle/_ref, built from /root/reference) timed on this
                 host's cores on a bounded sample of the same workload (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(n_workload: int):
    """Reference M4RI on the host cores, bounded sample: mzd_mul at n = 8192 and 16384 (sequential
    build) and mzd_mul_mp (OpenMP build, all cores) -- about 15-25 s of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_libs
    from m4ri_amd.mzd import Mzd
    ncpu = os.cpu_count() or 1
    ref = cpu_libs.reference()
    if ref is None:
        # no reference binary on this box: time our own plain-C restatement instead
        orc = cpu_libs.oracle()
        n = 4096
        A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
        t = time.perf_counter()
        orc.mul(None, A, B, 0)
        dt = time.perf_counter() - t
        return {"value": n ** 3 / dt, "unit": "bit-op/s", "cores": 1, "kind": "port",
                "sample": f"oracle gf2o_mul {n}^3, 1 run, {dt:.2f} s"}
    n = 16384
    A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
    best_seq = 1e30
    for _ in range(2):
        t = time.perf_counter()
        ref.mul(None, A, B, 0)
        best_seq = min(best_seq, time.perf_counter() - t)
    out = {"value": n ** 3 / best_seq, "unit": "bit-op/s", "cores": 1, "kind": "reference",
           "sample": f"reference mzd_mul(NULL,A,B,0) {n}^3 (1/{(n_workload // n) ** 3} of the workload's n^3), "
                     f"sequential SSE2 build, best of 2: {best_seq:.2f} s"}
    omp = cpu_libs.reference(openmp=True)
    if omp is not None and omp.has_mp:
        os.environ.setdefault("OMP_NUM_THREADS", str(ncpu))
        best_mp = 1e30
        for _ in range(2):
            t = time.perf_counter()
            omp.mul_mp(None, A, B, 0)
            best_mp = min(best_mp, time.perf_counter() - t)
        out["openmp"] = {"value": n ** 3 / best_mp, "cores": ncpu,
                         "sample": f"reference mzd_mul_mp {n}^3, OpenMP build, OMP_NUM_THREADS={ncpu}, best of 2: {best_mp:.2f} s"}
        if n ** 3 / best_mp > out["value"]:
            out["value"], out["cores"] = n ** 3 / best_mp, ncpu
            out["sample"] += f"; headline value = mzd_mul_mp on {ncpu} threads ({best_mp:.2f} s)"
    return out


LEAF_KERNELS = {1: "m4rm_leaf_kernel", 2: "m4rm7_kernel", 3: "m4rm8_kernel", 4: "m4rm8q_kernel"}
# (tile rows, tile columns, inner bits per stage, LDS-array clocks per stage): gathers at 256 B/clk/CU
# + table writes at 128 B/clk/CU, both measured with tools/ubench.hip
LEAF_LDS_MODEL = {2: (1024, 2048, 14, 2560), 3: (2048, 1024, 16, 2560), 4: (4096, 512, 32, 4608)}
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
    import torch
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


def leaf_traffic(n, world):
    """HBM bytes of one leaf launch from the rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950
    correction + WRITE_SIZE, MI355X_MICROARCH.md HBM section) -- counters cannot be read from inside
    this process, so the number is the one tools/prof_bench.sh measured for this exact command
    (profiles/leaf_traffic.json); null for any other configuration."""
    p = os.path.join(ROOT, "profiles", "leaf_traffic.json")
    if world != 1 or not os.path.exists(p):
        return None
    d = json.load(open(p))
    return d.get("bytes_per_launch") if d.get("n") == n else None


def bytes_sched(m, l, n, levels):
    """HBM bytes of the DECLARED, UNFUSED Strassen-Winograd schedule for C = A*B with
    `levels` levels -- every operand read once and every result written once per kernel: a leaf `mul`
    moves 8*(m*W(l) + l*W(n) + m*W(n)) bytes, a quadrant addition of r x c bits 3*8*r*W(c); per level 7
    products + 15 additions (4 on A-quadrant shape, 4 on B-quadrant shape, 7 on C-quadrant shape,
    strassen.c:111-150)."""
    W = lambda x: (x + 63) // 64
    if levels == 0:
        return 8 * (m * W(l) + l * W(n) + m * W(n))
    hm, hl, hn = m // 2, l // 2, n // 2
    adds = 3 * 8 * (4 * hm * W(hl) + 4 * hl * W(hn) + 7 * hm * W(hn))
    return adds + 7 * bytes_sched(hm, hl, hn, levels - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=65536, help="n of the n x n x n product")
    ap.add_argument("--workload", default="mul", choices=["mul", "leaf16384", "rect131072"],
                    help="mul: n^3 mzd_mul (the headline, configs[2]/[3]); leaf16384: configs[1]; "
                         "rect131072: 131072 x 8192 x 131072 (configs[4]), rows of A/C split over the ranks")
    ap.add_argument("--cutoff", type=int, default=0)
    ap.add_argument("--grid", default="", help="gi,gj,gh split of (m, n, l) over the ranks (default: sharding.default_grid)")
    ap.add_argument("--max-fuse", type=int, default=0, help="Strassen levels per fused pass (1..3; 0 = engine default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: development aid -- several ranks may share one GPU, P2P is staged through the host")
    ap.add_argument("--check", action="store_true",
                    help="after timing, every rank recomputes the full product on its own GPU and compares its owned block")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    m4ri_amd.init(dev)
    if args.max_fuse:
        m4ri_amd.set_max_fuse(args.max_fuse)
    stream = torch.cuda.current_stream().cuda_stream

    if args.workload == "leaf16384":
        n = 16384
    else:
        n = args.size
    assert n % 64 == 0
    M, L, N = (131072, 8192, 131072) if args.workload == "rect131072" else (n, n, n)
    wl, w = L // 64, N // 64  # words per row of A, and of B / C

    # ---- operands, resident in HBM (every rank generates the same A and B; it uses views) ----
    A = torch.empty((M, wl), dtype=torch.int64, device="cuda")
    B = torch.empty((L, w), dtype=torch.int64, device="cuda")
    seeds = (5, 6) if args.workload == "rect131072" else (3, 4)
    m4ri_amd.fill_dev(A.data_ptr(), wl, M, L, seeds[0], stream)
    m4ri_amd.fill_dev(B.data_ptr(), w, L, N, seeds[1], stream)
    grid = tuple(int(x) for x in args.grid.split(",")) if args.grid else ((world, 1, 1) if args.workload == "rect131072" else None)
    plan = sharding.make_plan(world, rank, M, L, N, grid=grid)
    r0, r1 = plan.row_range()
    c0, c1 = plan.col_range()
    k0, k1 = plan.inner_range()
    P = torch.empty((r1 - r0, (c1 - c0) // 64), dtype=torch.int64, device="cuda")  # this rank's block of C
    pw = P.shape[1]
    gh = plan.grid[2]
    cuts = sharding.ShardPlan._cuts(r1 - r0, gh, 1)
    recv_buf = torch.empty((cuts[plan.h + 1] - cuts[plan.h], pw), dtype=torch.int64, device="cuda") if gh > 1 else None

    def multiply(r0, r1, k0, k1, c0, c1):
        a_ptr = A.data_ptr() + 8 * (r0 * wl + k0 // 64)
        b_ptr = B.data_ptr() + 8 * (k0 * w + c0 // 64)
        if args.workload == "leaf16384":
            m4ri_amd.m4rm_dev(P.data_ptr(), pw, a_ptr, wl, b_ptr, w, r1 - r0, k1 - k0, c1 - c0, False, 0, stream)
        else:
            m4ri_amd.mul_dev(P.data_ptr(), pw, a_ptr, wl, b_ptr, w, r1 - r0, k1 - k0, c1 - c0, False, args.cutoff, stream)

    def send_recv(partner, send_rows, recv_rows):
        if args.backend == "gloo":  # host-staged (development only)
            out = P[send_rows[0]:send_rows[1]].cpu()
            inp = torch.empty(recv_buf.shape, dtype=torch.int64)
            reqs = [dist.isend(out, partner), dist.irecv(inp, partner)]
            for r in reqs:
                r.wait()
            recv_buf.copy_(inp)
            return recv_buf
        ops = [dist.P2POp(dist.isend, P[send_rows[0]:send_rows[1]], partner),
               dist.P2POp(dist.irecv, recv_buf, partner)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return recv_buf

    def xor_rows(rows, got):
        ptr = P.data_ptr() + 8 * rows[0] * pw
        m4ri_amd.xor_dev(ptr, pw, ptr, pw, got.data_ptr(), pw, rows[1] - rows[0], (c1 - c0), stream)

    def step():
        sharding.run_sharded(plan, multiply, xor_rows, send_recv)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    m4ri_amd.set_profiling(True)
    # per-step marks on the stream the products run on (no synchronisation inside the timed region)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        marks[k].record()
        step()
    marks[args.steps].record()
    fence()
    t1 = time.perf_counter()
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    stats = m4ri_amd.get_stats()  # last step's schedule + its leaf launch durations (HIP events)
    m4ri_amd.set_profiling(False)
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    ops = float(M) * L * N  # classical bit multiply-accumulates of the WHOLE product (AND+XOR = 1 op)
    value = ops * args.steps / elapsed

    if args.check:
        # recompute the whole product on this rank's GPU and compare the region this rank owns
        region = sharding.run_sharded(plan, multiply, xor_rows, send_recv)
        torch.cuda.synchronize()
        full = torch.empty((M, w), dtype=torch.int64, device="cuda")
        m4ri_amd.mul_dev(full.data_ptr(), w, A.data_ptr(), wl, B.data_ptr(), w, M, L, N, False, 0, stream)
        torch.cuda.synchronize()
        rr0, rr1, cc0, cc1 = region
        mine = P[rr0 - r0:rr1 - r0, :]
        ok = bool(torch.equal(mine, full[rr0:rr1, cc0 // 64:cc1 // 64]))
        print(f"[check] rank {rank} grid {plan.grid} owns rows {rr0}:{rr1} cols {cc0}:{cc1} -> {'OK' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            raise SystemExit(3)
        m4ri_amd.set_profiling(True)
        step()
        torch.cuda.synchronize()
        stats = m4ri_amd.get_stats()
        m4ri_amd.set_profiling(False)

    if rank == 0:
        leaf_launch_ms = stats.leaf_ms / max(1, stats.leaf_launches)
        leaf_launch_bytes = stats.leaf_bytes / max(1, stats.leaf_launches)
        achieved = leaf_launch_bytes / (leaf_launch_ms * 1e-3) / 1e9 if leaf_launch_ms > 0 else 0.0
        leaf_ops = float(stats.leaf_m) * stats.leaf_l * stats.leaf_n * stats.leaf_products
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
                "workload": (f"mzd_mul {n}x{n}x{n} (BASELINE.json configs[2]/[3]): Strassen-Winograd over M4RM leaves"
                             if args.workload == "mul" else
                             f"mzd_mul {M}x{L}x{N} (BASELINE.json configs[4]), rows of A/C over the ranks" if args.workload == "rect131072"
                             else f"mzd_mul_m4rm {n}^3 leaf only (BASELINE.json configs[1])"),
                "m": M, "l": L, "n": N,
                "ops_counted": "m*l*n bit multiply-accumulates (one AND+XOR = 1 op), classical count credited to Strassen",
                "input": "splitmix64 seeds 3 (A), 4 (B), uniform bits, resident in HBM",
                "grid": list(plan.grid),
                "per_rank_product": [r1 - r0, k1 - k0, c1 - c0],
                "strassen_levels": int(stats.levels),
                "leaf_shape": [int(stats.leaf_m), int(stats.leaf_l), int(stats.leaf_n)],
                "leaf_products_per_rank": int(stats.leaf_products),
                "workspace_GiB": stats.workspace_bytes / 2 ** 30,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": LEAF_KERNELS.get(int(stats.leaf_gen), "?") + " (the M4RM leaf; one batched launch per step, HIP events around that launch alone)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": leaf_traffic(n, world) if args.workload == "mul" else None,
                "launch_ms": leaf_launch_ms,
                "launches_per_step": int(stats.leaf_launches),
                "algorithmic_bytes_per_launch": leaf_launch_bytes,
                "leaf_bitops_per_sec": (leaf_ops / (stats.leaf_ms * 1e-3)) if stats.leaf_ms > 0 else 0.0,
                "aux_pass_bytes_per_step": stats.aux_bytes,
                "lds": lds_model(int(stats.leaf_gen), int(stats.leaf_m), int(stats.leaf_l), int(stats.leaf_n),
                                 int(stats.leaf_products) // max(1, int(stats.leaf_launches)), leaf_launch_ms),
                "note": "the leaf is LDS-bound by design (table gathers at 256 B/clk/CU), not HBM-bound: "
                        "its algorithmic HBM bytes are ~1e-3 of its LDS traffic, so frac is small; the `lds` object "
                        "prices the launch against the LDS-array cycles it needs (DESIGN.md 3.1)",
            },
        }
        out["step_ms_min"], out["step_ms_median"] = step_ms[0], step_ms[len(step_ms) // 2]
        if args.workload != "leaf16384":
            copy_gbs = measured_copy_gbs()
            # the whole product against the HBM roofline in SURVEY.md 8(d)'s terms (schedule bytes of
            # this rank's block product / step time); the compulsory bytes beside it
            pm, pl, pn = r1 - r0, k1 - k0, c1 - c0
            bs = float(bytes_sched(pm, pl, pn, int(stats.levels)))
            out["roofline_schedule"] = {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "bytes_sched_per_rank": bs, "levels": int(stats.levels),
                "bytes_compulsory_per_rank": float(bytes_sched(pm, pl, pn, 0)),
                "achieved": bs / (ms_per_step * 1e-3) / 1e9,
                "frac": bs / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "bytes_moved_by_our_fused_passes": stats.aux_bytes + stats.leaf_bytes,
                "copy_peak": copy_gbs, "frac_of_copy_peak": bs / (ms_per_step * 1e-3) / 1e9 / copy_gbs if copy_gbs else None,
                "note": "unfused reference schedule bytes (15 quadrant adds/level) over the measured step time; "
                        "our fused three-level passes move far fewer bytes; copy_peak = this GPU's measured "
                        "device-to-device copy rate (read + write bytes)",
            }
        if world == 1 and not args.no_cpu_baseline:
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
