#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X:

    GF(2) n^3-equivalent bit-ops/s and wall-clock of one n x n x n mzd_mul, n = 65536, at 1/2/4/8 GPUs (strong scaling).

A "step" is one whole product C = A*B on device-resident, synthetic (splitmix64, density 1/2) operands, everything through
libm4ri_amd.so's C ABI.  Inputs are in HBM before the timed region starts.
    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 65536] [--workload mul|leaf16384|rect131072]

N = 1: Strassen-Winograd levels over batched M4RM leaves on one GPU (m4ri_amd_mul_dev).
N > 1: the command that receives `--gpus N` is a CONTROLLER (under `torch.distributed.run` launcher rank 0 takes the role, the other
launcher ranks step aside).  It measures, each in processes of its own: the links (every ordered pair of GPUs, `config.links`), the
SAME product on ONE GPU (`speedup_vs_n1`), the product on N GPUs through a ladder of two transports, and the reference's multi-core
path on this host (`cpu_baseline`; the reference switches to its multi-core path inside the same command,
bench/bench_multiplication.c:94-103) -- and always ends with exactly one JSON line (`config.controller_wall_s`).
  transports  `peer` = ONE process driving all GPUs through libm4ri_amd.so's distributed matrices (m4ri_amd_dmat_mul,
              m4ri_amd/csrc/multi.hip: the schedules behind the C boundary -- what mzd_mul_mp runs -- pieces pulled by
              hipMemcpyPeerAsync on per-link copy streams, one host thread per GPU); `rccl` = one process per GPU, torch.distributed
              (backend nccl == RCCL) send/recv + all-gather.  `--transport auto`: the FIRST rung that completes is the line (peer,
              then rccl; with the development backend gloo: rccl, then peer) -- never the better of two; the other rung is measured
              briefly beside it (`config.transports_measured`) and in full only when its first 3 steps are within 1.3x.
  schedules   row slabs up to 4 ranks and for products a Strassen level cannot help (configs[4]); above, the sub-products of the top
              Strassen-Winograd level(s) over the ranks, slab-cyclic layout (m4ri_amd/sharding.py).
  numbers     `value` / `ms_per_step` = ONE product at a time (the metric is the wall clock of one mzd_mul); `pipelined_value` /
              `pipelined_ms_per_step` = a stream of independent products, two in flight (the transport of one under the
              multiplications of the other): a second, separately named measurement, never the headline.

Besides the contract fields the line carries "roofline" (the M4RM leaf launch against the HBM roofline: HIP events around every launch of
the timed steps; "traffic" from rocprofv3 PMC passes at N = 1), "roofline_schedule" (bytes of the declared, unfused
schedule over the step time against 8 TB/s), "cpu_baseline" (the real reference M4RI, oracle/_ref, on this host's cores) and "verified"
(SHA-256 of the C the timed steps produced against the reference's, tests/golden).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
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
from tools.bench_extras import host_api_timing, measure_leaf_traffic  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


# ---------------------------------------------------------------------------------------------------
def cpu_baseline_or_note(n_workload: int):
    """The reference M4RI itself on this host's cores (tests/cpu_baseline.py: config 1's timed region with the reference's stop rule,
    mzd_mul / mzd_mul_mp at 16384^3 and, when it fits ~75 s, mzd_mul_mp at the workload's own size on all cores)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from cpu_baseline import cpu_baseline
        return cpu_baseline(n_workload)
    except Exception as e:  # noqa: BLE001 -- the baseline is reported, never required for the GPU number
        return {"value": None, "unit": "bit-op/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}


# ---------------------------------------------------------------------------------------------------
# models and measurements around the number
# ---------------------------------------------------------------------------------------------------
LEAF_KERNELS = {1: "m4rm_leaf_kernel", 4: "m4rm8q_kernel", 5: "m4rm_small_kernel"}
# (tile rows, tile columns, inner bits per stage, LDS-array clocks per stage): gathers at 256 B/clk/CU
# + table writes at 128 B/clk/CU, both measured with tools/ubench.hip
LEAF_LDS_MODEL = {4: (4096, 512, 32, 4608)}
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


def pass_pattern_gbs(stream=0):
    """What this box's HBM gives the access pattern the Winograd passes have -- two streams read, one written, our own kernel
    (m4ri_amd_xor_dev on three 1 GiB operands): GB/s over read + written bytes, median of 5.  Reported beside the vendor peak as the
    practical ceiling of the passes; the headline fraction stays the one against 8 TB/s."""
    rows, w = 1 << 15, 1 << 12   # 32768 rows x 4096 words = 1 GiB per operand
    a, b, c = (torch.empty((rows, w), dtype=torch.int64, device="cuda") for _ in range(3))
    a.zero_()
    b.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for k in range(7):
        e0.record()
        m4ri_amd.xor_dev(c.data_ptr(), w, a.data_ptr(), w, b.data_ptr(), w, rows, w * 64, stream)
        e1.record()
        e1.synchronize()
        if k >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return 3.0 * rows * w * 8 / (ts[len(ts) // 2] * 1e-3) / 1e9


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


# ---------------------------------------------------------------------------------------------------
# the controller of an N > 1 run
# ---------------------------------------------------------------------------------------------------
LAUNCHER_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                "MASTER_ADDR", "MASTER_PORT", "NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_ASYNC_ERROR_HANDLING")
CONTROLLER_ONLY = {"--transport": 1, "--watchdog": 1, "--inner": 0, "--no-cpu-baseline": 0, "--no-links": 0, "--no-n1": 0}   # flag -> values that follow it


def child_env():
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV and not k.startswith("TORCHELASTIC_")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return env


def own_args(argv, drop=()):
    """This command's own arguments minus what the controller decides (and `drop`: flag -> number of values)."""
    skip_table = dict(CONTROLLER_ONLY, **dict(drop))
    args, skip = [], 0
    for a in argv:
        if skip:
            skip -= 1
        elif a in skip_table:
            skip = skip_table[a]
        elif not any(a.startswith(f + "=") for f in skip_table):
            args.append(a)
    return args


def run_child(cmd, watchdog_s, key='"metric"'):
    """One measurement in processes of its own (own session: on a timeout the whole group we started is killed, nothing else).
    Returns (the child's JSON line that contains `key` or None, why it failed or None, its other stdout lines)."""
    import signal
    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:
        proc = subprocess.Popen(cmd, env=child_env(), stdout=fo, stderr=fe, start_new_session=True)
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
    lines = [ln for ln in out_lines if ln.startswith("{") and key in ln]
    rest = [ln for ln in out_lines if not (ln.startswith("{") and key in ln)]
    if why is None and len(lines) != 1:
        why = f"{len(lines)} result lines"
    if why is not None:
        errs = [ln for ln in out_lines if ln.startswith("{") and '"error"' in ln]
        detail = errs[-1] if errs else " | ".join(err_tail.strip().splitlines()[-3:])
        return None, f"{why}: {detail}"[:700], rest
    return lines[0], None, rest


def run_rung(transport: str, n_ranks: int, argv: list, watchdog_s: float, extra=()):
    """One rung of the N > 1 ladder.  rccl: `torch.distributed.run --standalone` (it picks its own rendezvous port on 127.0.0.1) with
    one rank per GPU; peer: one process driving all GPUs."""
    me, args = os.path.abspath(__file__), own_args(argv)
    if transport == "rccl":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n_ranks}",
               me, *args, *extra, "--inner", "--transport", "rccl"]
    else:
        cmd = [sys.executable, me, *args, *extra, "--inner", "--transport", "peer"]
    return run_child(cmd, watchdog_s)


def controller(args) -> int:
    """`bench.py --gpus N` as a command: measure the links and the one-GPU time of the same product, start the ranks, watch them, fall
    down the ladder, time the reference's multi-core path, and end with exactly one JSON line either way."""
    t_start = time.perf_counter()
    real = args.backend == "nccl"
    ladder = {"auto": ["peer", "rccl"] if real else ["rccl", "peer"], "rccl": ["rccl"], "peer": ["peer"]}[args.transport]
    me = os.path.abspath(__file__)
    # (a) what the links give, before anything is scheduled over them: one process, every ordered pair of ranks
    links = None
    if not args.no_links:
        line, why, _ = run_child([sys.executable, me, "--gpus", str(args.gpus), "--inner", "--links-probe"] + (["--virtual-ranks"] if args.virtual_ranks else []),
                                 min(args.watchdog, 180.0), key='"links"')
        links = json.loads(line)["links"] if line else {"error": why}
    # (b) the same product of the same binary on ONE GPU, in the same invocation: the denominator of the speed-up
    n1 = None
    if not args.no_n1:
        line, why, _ = run_child([sys.executable, me, *own_args(sys.argv[1:], {"--gpus": 1, "--backend": 1, "--variant": 1, "--inflight": 1, "--overlap": 1,
                                                                            "--shard-levels": 1, "--check": 0, "--virtual-ranks": 0, "--slab-overlap": 1}),
                                  "--gpus", "1", "--no-cpu-baseline", "--no-api", "--no-traffic"], args.watchdog)
        n1 = json.loads(line) if line else {"error": why}
    # (c) the ladder: the first rung that completes is the line; the other is measured beside it, briefly unless it is close
    failed, done, primary = [], {}, None
    for transport in ladder:
        extra = ["--compare-ms", f"{done[primary]['ms_per_step']:.4f}"] if primary else []
        line, why, rest = run_rung(transport, args.gpus, sys.argv[1:], args.watchdog, extra)
        for ln in rest:
            print(ln, flush=True)
        if line is not None:
            done[transport] = json.loads(line)
            primary = primary or transport
            if not (args.transport == "auto" and real):
                break
        else:
            failed.append({"transport": transport, "reason": why})
    if not done:
        print(json.dumps({"error": f"no transport completed the {args.gpus}-GPU run", "gpus_requested": args.gpus, "transport_fallback": failed,
                          "links": links, "controller_wall_s": time.perf_counter() - t_start}), flush=True)
        return 1
    out = done[primary]
    cfg = out["config"]
    cfg["transport"] = primary
    cfg["transport_rule"] = "the first rung of the ladder that completes is the line (never the better of two); the other rung is in transports_measured"
    cfg["transport_fallback"] = [f for f in failed if ladder.index(f["transport"]) < ladder.index(primary)]
    later = [f for f in failed if ladder.index(f["transport"]) > ladder.index(primary)]
    if later:
        cfg["transports_unavailable"] = later
    if len(done) > 1:
        cfg["transports_measured"] = {t: {k: d.get(k) for k in ("value", "ms_per_step", "pipelined_value", "pipelined_ms_per_step", "host_issue_ms_per_step",
                                                              "steps", "stopped_early")} for t, d in done.items()}
    if links is not None:
        cfg["links"] = links
    if n1 is not None:
        if "ms_per_step" in n1:
            out["speedup_vs_n1"] = {"n1_ms_per_step": n1["ms_per_step"], "n1_value": n1["value"], "one_product": n1["ms_per_step"] / out["ms_per_step"],
                                    "pipelined": (n1["ms_per_step"] / out["pipelined_ms_per_step"]) if out.get("pipelined_ms_per_step") else None,
                                    "what": "the same product through the same binary on ONE GPU, measured by this command before the N-GPU run "
                                            f"({n1['steps']} steps); the driver computes scaling efficiency itself from its own per-N runs"}
        else:
            out["speedup_vs_n1"] = n1
    # (d) the reference's multi-core path on this host's cores, once, outside the ranks (they are gone by now)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_or_note(cfg["m"] if cfg.get("m") and cfg.get("m") == cfg.get("n") == cfg.get("l") else args.size)
    cfg["controller_wall_s"] = time.perf_counter() - t_start
    print(json.dumps(out), flush=True)
    return 0


def links_probe(args) -> int:
    """`--links-probe`: one process, all GPUs -- every ordered pair copies 256 MiB one pair at a time and all pairs at once over the
    library's own link streams (m4ri_amd_multi_link_probe); peer access pair by pair."""
    ndev = m4ri_amd.lib().m4ri_amd_device_count()
    if ndev < args.gpus and not args.virtual_ranks:
        print(json.dumps({"links": {"error": f"only {ndev} device(s) visible for {args.gpus} ranks"}}), flush=True)
        return 0
    m4ri_amd.set_devices([i % max(1, ndev) for i in range(args.gpus)])
    p = m4ri_amd.multi_link_probe(256 << 20)
    p["what"] = ("hipMemcpyPeerAsync of 256 MiB per ordered pair of ranks on the library's link streams: one pair at a time, then all pairs at once"
                 + ("; RANKS SHARE DEVICES here (virtual ranks): these are on-device blit rates, not link rates" if p["ranks_share_devices"] else ""))
    print(json.dumps({"links": p}), flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=65536, help="n of the n x n x n product")
    ap.add_argument("--workload", default="mul", choices=["mul", "leaf16384", "rect131072"],
                    help="mul: n^3 mzd_mul (the headline, configs[2]/[3]); leaf16384: configs[1]; rect131072: 131072 x 8192 x 131072 (configs[4])")
    ap.add_argument("--cutoff", type=int, default=0)
    ap.add_argument("--variant", default="auto", choices=["auto", "strassen", "slabs"],
                    help="N > 1: what is handed out to the ranks (auto: slabs up to 4 ranks and for thin products, strassen above)")
    ap.add_argument("--shard-levels", type=int, default=0, help="strassen variant: sharded levels (1, 2; 0 = automatic)")
    ap.add_argument("--max-fuse", type=int, default=0, help="Strassen levels per fused pass (1..4; 0 = engine default: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the host-API (PCIe-inclusive) timing")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--no-verify", action="store_true", help="skip the SHA-256 check of C against the reference's")
    ap.add_argument("--no-links", action="store_true", help="N > 1 controller: skip the link probe")
    ap.add_argument("--no-n1", action="store_true", help="N > 1 controller: skip the one-GPU run of the same product")
    ap.add_argument("--probe", action="store_true", help="internal: one warm-up + one product, nothing else (run under rocprofv3)")
    ap.add_argument("--links-probe", action="store_true", help="internal: measure the links between the ranks' devices and print them")
    ap.add_argument("--compare-ms", type=float, default=0.0,
                    help="internal (second rung of the ladder): ms per step of the rung that is already the line -- after 3 steps slower than 1.3x this, "
                         "stop and report those 3 steps")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: development aid -- several ranks may share one GPU, pieces are staged through the host")
    ap.add_argument("--check", action="store_true",
                    help="after timing, every rank recomputes the full product on its own GPU and compares the part of C it holds")
    ap.add_argument("--seeds", default="", help="a,b: splitmix64 seeds of A and B (default 3,4; 5,6 for rect131072)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: initialise the process group anyway and run the N > 1 code path (its collectives, "
                         "batches and fences) at world size 1 -- how a one-GPU box puts the RCCL path through its paces")
    ap.add_argument("--inflight", type=int, default=0, choices=[0, 1, 2],
                    help="N > 1: 2 (the default there) = after the timed loop (always one product at a time: `value`) also time a stream of products, "
                         "two in flight -- reported as pipelined_value / pipelined_ms_per_step; 1 = skip that")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "peer"],
                    help="N > 1: peer = one process, all GPUs, m4ri_amd_dmat_mul (peer copies); rccl = one process per GPU over torch.distributed; "
                         "auto = the first of (peer, rccl) that completes within --watchdog seconds")
    ap.add_argument("--watchdog", type=float, default=240.0, help="N > 1: seconds one rung of the transport ladder may take before its processes are killed")
    ap.add_argument("--virtual-ranks", action="store_true",
                    help="peer transport: allow more ranks than visible GPUs (ranks share devices round robin: how a one-GPU box tests the path)")
    ap.add_argument("--inner", action="store_true", help="internal: this process is a rank / the single process of a rung the controller started")
    ap.add_argument("--slab-overlap", type=int, default=-1,
                    help="slabs variant: multiply with the rank's own slab of B while the all-gather of the others runs (1 / 0; default: on up to 4 ranks)")
    ap.add_argument("--dims", default="", help="m,l,n of a general product (overrides --size; ragged sizes exercise the uneven slabs)")
    ap.add_argument("--overlap", default="",
                    help="strassen variant: R or RxC -- row (x column) chunks per sub-product whose transport overlaps the products "
                         "(1 = none; default: 2 when a rank owns one large sub-product, else 1)")
    args = ap.parse_args()

    if args.links_probe:
        raise SystemExit(links_probe(args))
    if args.gpus > 1 and not args.inner and not args.probe:
        # this command is the controller of an N-GPU run.  Under a launcher that already started N copies of it, copy 0 takes the
        # role and the others step aside: the ranks that do the work are the controller's own, watched and replaceable
        if int(os.environ.get("RANK", "0")) != 0:
            return
        raise SystemExit(controller(args))
    peer = args.transport == "peer" and args.gpus > 1
    world, rank, local_rank = (args.gpus, 0, 0) if peer else (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
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
    pipelined_step = None         # peer transport: product k of a stream, issued on lane k & 1
    last = {"slot": 0}            # the slot that holds the C of the last product
    inflight = (args.inflight or 2) if multi else 1

    # ---------------------------------------------------------------- N == 1 -----------------------
    if not multi:
        A = torch.empty((M, wl), dtype=torch.int64, device="cuda")
        B = torch.empty((L, w), dtype=torch.int64, device="cuda")
        m4ri_amd.fill_dev(A.data_ptr(), wl, M, L, seeds[0], stream)
        m4ri_amd.fill_dev(B.data_ptr(), w, L, N, seeds[1], stream)
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
        v = m4ri_amd.VARIANT_SLABS if args.variant == "slabs" else m4ri_amd.VARIANT_STRASSEN
        if v == m4ri_amd.VARIANT_STRASSEN:
            lay = m4ri_amd.LAYOUT_CYCLIC2 if plan.levels == 2 else m4ri_amd.LAYOUT_CYCLIC1
        else:
            lay = m4ri_amd.LAYOUT_ROWS
        dA, dB = m4ri_amd.Dmat(M, L, lay).fill(seeds[0]), m4ri_amd.Dmat(L, N, lay).fill(seeds[1])
        dCs = [m4ri_amd.Dmat(M, N, lay) for _ in range(inflight)]
        dC = dCs[0]
        issue_s = [0.0]

        def step():   # asynchronous: returns when every rank's host thread has issued its part (the devices work on); one product at a time:
            ta = time.perf_counter()   # every product of lane 0 starts behind the previous one on every rank
            m4ri_amd.dmat_mul(dC, dA, dB, False, args.cutoff, v)
            issue_s[0] += time.perf_counter() - ta

        def pipelined_step(k):   # a stream of independent products, alternating lanes: two in flight
            m4ri_amd.dmat_mul(dCs[k % inflight], dA, dB, False, args.cutoff, v, lane=k % inflight)
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
        Bs = torch.zeros((kb, w), dtype=torch.int64, device="cuda")   # the slabs are where the inputs live
        if mr:
            m4ri_amd.fill_rows_dev(As.data_ptr(), wl, rc[rank], mr, L, seeds[0], stream)
        if lr:
            m4ri_amd.fill_rows_dev(Bs.data_ptr(), w, bc[rank], lr, N, seeds[1], stream)
        # the all-gather of B under the first product: needs every slab boundary of B on a word of A's rows, and pays when a rank's
        # own slab is a large part of the inner dimension (few ranks); --slab-overlap 0/1 overrides
        aligned = all(c % 64 == 0 for c in bc[:-1]) and lr > 0
        slab_overlap = aligned and (world <= 4 if args.slab_overlap < 0 else bool(args.slab_overlap))

        def step():   # one product at a time, always on slot 0
            Cs, Bfull = Cs_slots[0], Bfull_slots[0]
            if slab_overlap:
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
        config_extra.update({"parallelism": f"row slabs x{world} + all-gather of B", "variant": "slabs", "layout": "distributed",
                             "slab_rows": [ka, kb], "all_gather_under_first_product": bool(slab_overlap),
                             "bytes_over_links_per_step": 8 * kb * w * (world - 1), "collective": "all_gather_into_tensor(B)"})

    # ---------------------------------------------------------------- N > 1, Strassen sub-products --
    else:
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
        group = m4ri_amd.shard_group(plan, args.cutoff)   # a rank's sub-products in batched products of this many (whole sub-products only)
        for b, (g0, rows) in enumerate(runs_a):   # the slabs are where the inputs live: fill them straight from the streams
            m4ri_amd.fill_rows_dev(bufs["local_a"].data_ptr() + 8 * b * sa * wl, wl, g0, rows, L, seeds[0], stream)
        for b, (g0, rows) in enumerate(runs_b):
            m4ri_amd.fill_rows_dev(bufs["local_b"].data_ptr() + 8 * b * sb * w, w, g0, rows, N, seeds[1], stream)

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

            def do_product_group(q0, count):   # `count` of the rank's sub-products, rounds q0 .., as one batched product
                m4ri_amd.mul_batch_dev(sb_["prod"].data_ptr() + 8 * q0 * plan.bm * plan.cwn, plan.cwn, plan.bm * plan.cwn,
                                       sb_["oper_a"].data_ptr() + 8 * q0 * plan.bm * plan.cwl, plan.cwl, plan.bm * plan.cwl,
                                       sb_["oper_b"].data_ptr() + 8 * q0 * plan.bl * plan.cwn, plan.cwn, plan.bl * plan.cwn,
                                       plan.bm, plan.bl, plan.cwn * 64, count, False, args.cutoff, stream)

            def do_up():
                m4ri_amd.shard_up_dev(plan, rank, sb_["slabs_p"].data_ptr(), sb_["local_c"].data_ptr(), w, False, stream)
            return sharding.StrassenShardedStep(plan, rank, sb_, do_down, do_product, do_up, exchange, lambda d, s: d.copy_(s), chunks=chunks_,
                                                group=group if tuple(chunks_) == (1, 1) else 1, product_group=do_product_group)
        # one product at a time (step(), `value`): row chunks hide part of its own transport.  Two products in flight: the neighbours'
        # multiplications hide all of it, so the sub-products stay whole (their halves cost up to 3 % more than the whole)
        single_step = sharded_step(slot_bufs[0], chunks)
        chunks_loop = chunks if (args.overlap or inflight == 1) else (1, 1)
        if inflight > 1:
            phase_steps = [sharded_step(sb_, chunks_loop) for sb_ in slot_bufs]

        def step():
            single_step.start()
            single_step.multiply()
            single_step.finish()
        per_rank_product = [plan.bm, plan.bl, plan.cwn * 64]
        moved = sum(pc.words * 8 for side, j, r, pc in sharding.strassen_pieces(plan, (0, 1, 2)) if pc.holder != pc.owner)
        config_extra.update({"parallelism": f"strassen-sharded x{world}", "variant": "strassen", "layout": "distributed", "sharded_levels": plan.levels,
                             "sub_products": plan.nprod, "sub_products_on_busiest_rank": len(sharding.owned_products(plan, 0)),
                             "bytes_over_links_per_step": moved, "links_used": world * (world - 1), "overlap_chunks": list(chunks),
                             "overlap_chunks_in_the_pipelined_loop": list(chunks_loop), "sub_products_per_batched_product": group if tuple(chunks) == (1, 1) else 1,
                             "collective": (f"batched isend/irecv, one batch per group of {group} round(s): the group's operands out, its products back "
                                            f"({-(-(-(-plan.nprod // world)) // group)} + {-(-(-(-plan.nprod // world)) // group)} batches per product)"
                                            if group > 1 and tuple(chunks) == (1, 1) else
                                            f"batched isend/irecv (one group per batch: operands out per round and row / column chunk, products back "
                                            f"per unit; {len(sharding.chunk_bounds(plan, chunks[0]))} x {len(sharding.column_bounds(plan, chunks[1]))} unit(s) "
                                            f"x {-(-plan.nprod // world)} round(s))")})
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
    def run_steps(count, marks=None, pipelined=False):
        """`count` products back to back: one at a time through step(), or -- pipelined -- two in flight over two buffer slots / lanes."""
        before = (lambda k: marks[k].record()) if marks is not None else None
        if not pipelined:
            for k in range(count):
                if before is not None:
                    before(k)
                step()
            last["slot"] = 0
        elif peer:
            for k in range(count):
                pipelined_step(k)
            last["slot"] = (count - 1) % inflight
        else:
            sharding.run_products(lambda k: phase_steps[k % inflight], count, inflight, before)
            last["slot"] = (count - 1) % inflight

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cuda")
        if args.backend == "gloo":
            tt = tt.cpu()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed(count, pipelined=False, marks=None):
        fence()
        if peer:
            issue_s[0] = 0.0
        t0 = time.perf_counter()
        run_steps(count, marks, pipelined)
        if marks is not None:
            marks[count].record()
        t_posted = time.perf_counter()   # every step is posted; the devices may still be working
        fence()
        return max_over_ranks(time.perf_counter() - t0), t_posted - t0

    run_steps(args.warmup)
    stopped_early = None
    steps = args.steps
    if args.compare_ms > 0 and args.steps > 3:
        # the second rung of the ladder: the line exists already.  Three steps first; a rung that is clearly slower stops there
        quick, _ = timed(3)
        if 1e3 * quick / 3 > 1.3 * args.compare_ms:
            stopped_early = {"after_steps": 3, "ms_per_step": 1e3 * quick / 3, "compare_ms": args.compare_ms, "rule": "> 1.3 x the rung that is the line"}
            steps = 3
    m4ri_amd.set_profiling(2)  # leaf launches bracketed by HIP events on their stream, accumulated over all steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    elapsed, posted = timed(steps, marks=marks)
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps)) if not peer else None
    stats = m4ri_amd.get_stats()  # last product's schedule + the leaf launch durations of ALL timed steps
    m4ri_amd.set_profiling(0)
    # how long the HOST needs to post one step, no fence inside: must stay well below the step for the devices never to wait for it
    host_issue_ms = 1e3 * max_over_ranks((issue_s[0] if peer else posted) / steps)
    ms_per_step = 1e3 * elapsed / steps
    ops = float(M) * L * N  # classical bit multiply-accumulates of the WHOLE product (AND+XOR = 1 op)
    value = ops * steps / elapsed
    pipelined = None
    if multi and inflight > 1 and stopped_early is None and (phase_steps is not None or peer):
        # a second, separately named measurement: the throughput of a STREAM of independent products, two in flight (the transport of
        # the neighbouring products under the multiplications of the current one).  Never the headline: the metric is one mzd_mul
        run_steps(2, None, True)
        tp, _ = timed(steps, pipelined=True)
        pipelined = {"pipelined_value": ops * steps / tp, "pipelined_ms_per_step": 1e3 * tp / steps}
    if multi and not peer and args.variant == "slabs":
        Cs, Bfull = Cs_slots[last["slot"]], Bfull_slots[last["slot"]]
    if multi and not peer and args.variant == "strassen":
        bufs = slot_bufs[last["slot"]]
    if peer:
        dC = dCs[last["slot"]]

    # ---------------------------------------------------------------- correctness of what was timed --
    verified, Chost = None, None
    want = golden_sha("mul", M, L, N, seeds) if (not args.no_verify and args.workload != "leaf16384") else None
    if want is not None:
        what = "C of the last timed step vs the real reference's product of the same inputs (tests/golden/sha256.json)"
        got = None
        if not multi:
            got = sha_of_device_rows(Cfull)
        elif peer:
            Chost = dC.download()   # every device its own rows, outside the timed region
            got = hashlib.sha256(Chost.masked().tobytes()).hexdigest()
            what = f"C of the last product, downloaded from the {world} ranks, vs the real reference's product of the same inputs (tests/golden/sha256.json)"
        else:
            # gather the ranks' rows of C on rank 0 (outside the timed region, over the same transport)
            if rank == 0:
                Cfull = torch.empty((M, w), dtype=torch.int64, device="cuda")

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
                what = f"C of the last product, gathered from the {world} ranks, vs the real reference's product of the same inputs (tests/golden/sha256.json)"
        if got is not None:
            verified = {"sha256": got, "matches_reference": got == want, "what": what}
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
            ok, what = True, f"the whole C of {len(dCs)} lane(s), downloaded"
            for k, d in enumerate(dCs if pipelined is not None else dCs[:1]):
                Ck = Chost if (d is dC and Chost is not None) else d.download()
                ok = ok and bool(np.array_equal(Ck.valid_words().view(np.int64), full.cpu().numpy()))
        elif args.variant == "slabs":
            ok = bool(torch.equal(Cs, full[rc[rank]:rc[rank + 1]]))
            what = f"rows {rc[rank]}:{rc[rank + 1]} of C"
        else:
            ok = True
            for b, (g0, rows) in enumerate(runs_a):
                ok = ok and bool(torch.equal(bufs["local_c"][b * sa * w:(b * sa + rows) * w].reshape(rows, w), full[g0:g0 + rows]))
            what = f"slabs {[(g0, g0 + rows) for g0, rows in runs_a]} of C"
        if rank == 0 and Cfull is not None and not peer:
            ok = ok and bool(torch.equal(Cfull, full))
            what += " + the gathered C"
        print(f"[check] rank {rank} {config_extra.get('parallelism')} {what} -> {'OK' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            raise SystemExit(3)

    if rank == 0:
        # N > 1, Strassen sub-products: the engine's stats are those of rank 0's LAST call -- one sub-product, or a batched product of the
        # last group of its sub-products; per rank and product there are `calls_rank` such calls
        owned0 = len(sharding.owned_products(plan, 0)) if (multi and args.variant == "strassen") else 1
        grp0 = (config_extra.get("sub_products_per_batched_product", 1) if not peer else int(m4ri_amd.multi_stats().group)) if (multi and args.variant == "strassen") else 1
        grp0 = max(1, grp0)
        last_batch = (owned0 % grp0 or grp0) if owned0 else 1
        sub_leaves = int(stats.leaf_products) // max(1, last_batch)   # leaf products of ONE sub-product (of the whole product at N = 1)
        launches = max(1, int(stats.cum_leaf_launches))
        leaf_launch_ms = stats.cum_leaf_ms / launches               # mean over every leaf launch of the timed steps
        leaf_launch_bytes = stats.leaf_bytes / max(1, stats.leaf_launches)
        achieved = leaf_launch_bytes / (leaf_launch_ms * 1e-3) / 1e9 if leaf_launch_ms > 0 else 0.0
        leaf_ops = float(stats.leaf_m) * stats.leaf_l * stats.leaf_n * stats.leaf_products
        leaf_ms_per_product = leaf_launch_ms * max(1, int(stats.leaf_launches))
        traffic, traffic_detail = None, "not measured for this configuration"
        if not multi and args.workload == "mul" and not args.no_traffic:
            traffic, traffic_detail = measure_leaf_traffic(n, args.cutoff)
        workload = (f"mzd_mul_m4rm {n}^3 leaf only (BASELINE.json configs[1])" if args.workload == "leaf16384" else
                    f"mzd_mul {M}x{L}x{N} (BASELINE.json configs[4])" if args.workload == "rect131072" else
                    f"mzd_mul {n}x{n}x{n} (BASELINE.json configs[2]/[3]): Strassen-Winograd over M4RM leaves" if (M, L, N) == (65536, 65536, 65536) else
                    f"mzd_mul {M}x{L}x{N} (not a BASELINE.json configuration): Strassen-Winograd over M4RM leaves")
        if int(stats.levels) >= 2 and int(stats.leaf_products) % 7 ** int(stats.levels):   # fewer leaf products than 7 per level
            workload = workload.replace("Strassen-Winograd over", "Strassen-Winograd levels, the fused bottom ones by a rank-47 scheme of the 4 x 4 x 4 block product, over")
        out = {
            "metric": "gf2_matmul_n3_equiv_bitops_per_sec", "value": value, "unit": "bit-op/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": workload, "m": M, "l": L, "n": N,
                "ops_counted": "m*l*n bit multiply-accumulates (one AND+XOR = 1 op), classical count credited to Strassen",
                "input": f"splitmix64 seeds {seeds[0]} (A), {seeds[1]} (B), uniform bits, resident in HBM" + (" (slab-cyclic over the ranks)" if multi else ""),
                "per_rank_product": per_rank_product,
                "strassen_levels": int(stats.levels),
                # the engine's own plan for the per-rank product: rows in blocks [rows, levels], largest first (one block = one product)
                "row_blocks": ([list(b) for b in m4ri_amd.plan_row_blocks(*per_rank_product)] if args.workload != "leaf16384" and not args.cutoff else None),
                "leaf_shape": [int(stats.leaf_m), int(stats.leaf_l), int(stats.leaf_n)],
                "leaf_products_per_rank": sub_leaves * owned0,
                "workspace_GiB": stats.workspace_bytes / 2 ** 30,
                **config_extra,
            },
            "roofline": {
                # the resource that binds the kernel is the LDS array (`lds` below); achieved / peak / frac are the HBM figures SURVEY.md 8(d) asks for
                "bound": "lds",
                "kernel": LEAF_KERNELS.get(int(stats.leaf_gen), "?") + " (the M4RM leaf; HIP events around every launch on its stream, mean over all timed steps)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_detail": traffic_detail,
                "launch_ms": leaf_launch_ms, "launches_timed": int(stats.cum_leaf_launches), "launches_per_product": int(stats.leaf_launches),
                "algorithmic_bytes_per_launch": leaf_launch_bytes,
                "leaf_bitops_per_sec": (leaf_ops / (leaf_ms_per_product * 1e-3)) if leaf_ms_per_product > 0 else 0.0,
                "aux_pass_bytes_per_product": stats.aux_bytes,
                "lds": lds_model(int(stats.leaf_gen), int(stats.leaf_m), int(stats.leaf_l), int(stats.leaf_n),
                                 int(stats.leaf_products) // max(1, int(stats.leaf_launches)), leaf_launch_ms),
                "note": "the leaf is LDS-bound by design (table gathers at 256 B/clk/CU), not HBM-bound: its algorithmic HBM bytes are ~1e-3 of its "
                        "LDS traffic, so frac is small; the `lds` object prices the launch against the LDS-array cycles it needs (DESIGN.md 3.1)",
            },
        }
        lds = out["roofline"]["lds"]
        if lds and isinstance(traffic_detail, dict) and traffic_detail.get("gui_active_cycles"):
            # in cycles the clock drops out: LDS-array cycles the launch needs / cycles it took (same launch, profiled pass)
            need = lds["bound_ms"] * 1e-3 * PEAK_CLOCK_HZ
            lds["frac_in_cycles"] = need / traffic_detail["gui_active_cycles"]
            lds["effective_clock_hz"] = traffic_detail["effective_clock_hz"]
            lds["note"] = ("frac prices the launch at the 2.4 GHz peak clock, frac_in_cycles at the clock it really ran at (GRBM_GUI_ACTIVE / duration): "
                           "the kernel sits on the chip's power limit (measured: 1338 W of a 1400 W socket cap, the firmware's package-power limiter active 82 % of "
                           "the launch's time, no thermal limiter -- profiles/r04_leaf_power/).  Of the cycles with the LDS idle ~10 % are bubbles of the gather "
                           "pipeline itself (gathers alone: 90 % busy) and ~4.5 % the stage barrier (profiles/r03_leaf_decomposition/README.md)")
        if step_ms:
            out["step_ms_min"], out["step_ms_median"] = step_ms[0], step_ms[len(step_ms) // 2]
        out["host_issue_ms_per_step"] = host_issue_ms
        if stopped_early is not None:
            out["stopped_early"] = stopped_early
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
                                                         "row_chunks": ms.chunks, "sub_products_per_batched_product": ms.group, "gather_under_first_product": bool(ms.overlap),
                                                         "operands_converted": ms.converted, "bytes_over_links_per_step": ms.link_bytes,
                                                         "rank_pairs_copying_through_the_host": ms.pairs_staged},
                                      "timeline_ms_last_lane0_product": {str(r): m4ri_amd.multi_timeline(r) for r in range(world)},
                                      "timeline_marks": ("strassen: down pass done; per row chunk: operands in, product done; result slabs in; up pass done"
                                                         if ms.variant == m4ri_amd.VARIANT_STRASSEN else "slabs: gather done; first product done; all done")})
        if verified is not None:
            out["verified"] = verified
        if args.workload != "leaf16384":
            # the whole product against the HBM roofline in SURVEY.md 8(d)'s terms (schedule bytes of this rank's products / step time)
            pm, pl, pn = per_rank_product
            nprod_rank = 1 if (not multi or args.variant != "strassen") else len(sharding.owned_products(plan, 0))
            if peer and args.variant == "strassen":   # the library's own padding (every dimension to 256 bits) and row chunks
                pm, pl, pn = plan.bm, plan.bl, plan.cwn * 64
            bs = float(bytes_sched(pm, pl, pn, int(stats.levels))) * nprod_rank
            ceiling = pass_pattern_gbs(stream)
            frac = bs / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline_schedule"] = {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_sched_per_rank": bs, "levels": int(stats.levels),
                "bytes_compulsory_per_rank": float(bytes_sched(pm, pl, pn, 0)) * nprod_rank,
                "achieved": bs / (ms_per_step * 1e-3) / 1e9, "frac": frac,
                "bytes_moved_by_our_fused_passes": stats.aux_bytes + stats.leaf_bytes,
                # four fused levels = two applications of a rank-R 4 x 4 x 4 scheme (R^2 leaves, not 7^4); bytes_sched stays the REFERENCE's schedule
                "leaf_products_run": sub_leaves * nprod_rank, "leaf_products_declared": 7 ** int(stats.levels) * nprod_rank,
                "pass_pattern_peak": ceiling,
                "north_star_60pct": bool(frac >= 0.6),
                "note": "frac = unfused reference schedule bytes (15 quadrant adds/level, SURVEY.md 8(d)) of the rank's sub-product(s) over the measured step "
                        "time, against the 8 TB/s peak: the figure north_star's 60 % is asked of.  It is not met and cannot be by this design: 89 % of the step "
                        "is the M4RM leaf, which is LDS- and power-bound, not HBM-bound (SURVEY.md 8(d) predicted exactly that; DESIGN.md 3.1), and the depth "
                        "is chosen to minimise wall clock, never to inflate this fraction.  pass_pattern_peak = what this box's HBM gives our own two-reads-"
                        "one-write kernel (median of 5), the practical ceiling of the passes; our fused multi-level passes move far fewer bytes than the "
                        "unfused schedule (no fraction is quoted against a per-box ceiling: it would move with the box)",
            }
        if not multi and args.workload != "leaf16384" and not args.no_api:
            try:
                out["api"] = host_api_timing(A, B, M, L, N, args.cutoff)
                out["api_ms"] = out["api"]["c_given_ms_min"]
            except Exception as e:  # noqa: BLE001 -- reported beside the number, never required for it
                out["api"] = {"error": repr(e)}
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_or_note(n)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
