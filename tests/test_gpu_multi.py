"""One product over several ranks on the GPU (include/m4ri_amd.h part 4; the multi-device meaning of the
reference's mzd_mul_mp / mzd_addmul_mp, m4ri/mp.c:158-324).  A one-GPU box runs the path with several
"virtual" ranks on device 0: every rank has its own stream, buffers and slabs, pieces move by peer copies
(device 0 to itself), sub-products run through the one engine of the device -- the same code a node of 8
GPUs runs, minus the links.  Checked bit for bit against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)
    yield
    m4ri_amd.set_devices([])
    m4ri_amd.set_multi_threshold(16384)


SHAPES = [(64, 128, 128), (100, 256, 256), (77, 130, 65), (203, 300, 257), (1025, 1025, 1025), (2048, 2048, 4096),
          (1, 64, 64), (5, 1, 3), (1710, 1290, 1000), (4096, 3528, 4096)]


@pytest.mark.parametrize("world,levels", [(2, 1), (2, 2), (3, 0), (4, 2), (8, 1), (8, 2)])
def test_mul_multi_virtual_ranks(oracle, world, levels):
    m4ri_amd.set_devices([0] * world)
    for (m, l, n) in SHAPES:
        A, B = Mzd.random(m, l, 11), Mzd.random(l, n, 12)
        want = oracle.mul(None, A, B, 0)
        C = Mzd.random(m, n, 13)  # dirty C is overwritten
        assert m4ri_amd.mul_multi(C, A, B, False, 0, levels).equal(want), (world, levels, m, l, n)
        assert not (C.valid_words()[:, -1] & ~np.uint64(C.high_bitmask)).any()
        C0 = Mzd.random(m, n, 14)
        want2 = oracle.addmul(C0.copy(), A, B, 0)
        assert m4ri_amd.mul_multi(C0, A, B, True, 0, levels).equal(want2), ("addmul", world, levels, m, l, n)


def test_mzd_mul_mp_spreads_over_the_configured_devices(oracle):
    """The reference-named entry points take the multi-device path once the product is large enough."""
    m4ri_amd.set_devices([0, 0, 0, 0])
    old = m4ri_amd.set_multi_threshold(512)
    try:
        for (m, l, n, cutoff) in [(1500, 2000, 1700, 0), (4096, 4096, 4096, 1024), (600, 513, 700, 0), (3, 2000, 2000, 0)]:
            A, B = Mzd.random(m, l, 21), Mzd.random(l, n, 22)
            want = oracle.mul(None, A, B, 0)
            assert m4ri_amd.mzd_mul_mp(None, A, B, cutoff).equal(want)
            C0 = Mzd.random(m, n, 23)
            want2 = oracle.addmul(C0.copy(), A, B, 0)
            assert m4ri_amd.mzd_addmul_mp(C0, A, B, cutoff).equal(want2)
        A = Mzd.random(1024, 1024, 24)
        assert m4ri_amd.mzd_mul_mp(None, A, A, 0).equal(oracle.mul(None, A, A, 0))  # A == B
    finally:
        m4ri_amd.set_multi_threshold(old)


def test_windows_with_excess_keep_their_parents(oracle):
    """Operands and result are windows with non-zero excess inside pattern-filled parents
    (tests/test_smallops.c:115-121): every bit of C's parent outside the window survives."""
    m4ri_amd.set_devices([0, 0, 0])
    for (M, N, m, n) in [(1024, 1024, 513, 511), (1024, 1024, 512, 798), (2048, 2048, 1024, 1024)]:
        PA, PB, PC = Mzd.random(M, N, 31), Mzd.random(M, N, 32), Mzd.random(M, N, 33)
        a, b, c = PA.window(0, 0, m, n), PB.window(0, 64, n, 64 + m), PC.window(3, 128, 3 + m, 128 + m)
        PCo = Mzd(M, N, buf=PC.buf.copy())
        co = PCo.window(3, 128, 3 + m, 128 + m)
        oracle.mul(co, a.copy(), b.copy(), 0)
        m4ri_amd.mul_multi(c, a, b, False, 0, 1)
        assert np.array_equal(PC.buf, PCo.buf)
        oracle.addmul(co, a.copy(), b.copy(), 0)
        m4ri_amd.mul_multi(c, a, b, True, 0, 2)
        assert np.array_equal(PC.buf, PCo.buf)


def test_large_product_matches_single_device(oracle):
    """16384^3 over 8 ranks (7 sub-products of 8192^3) and over 4 ranks (49 of 4096^3) == the one-GPU product."""
    n = 16384
    A, B = Mzd.random(n, n, 41), Mzd.random(n, n, 42)
    ref = m4ri_amd.mzd_mul(None, A, B, 0)
    for world in (8, 4):
        m4ri_amd.set_devices([0] * world)
        assert m4ri_amd.mul_multi(Mzd.init(n, n), A, B, False, 0, 0).equal(ref)


@pytest.mark.parametrize("world,variant", [(2, "strassen"), (4, "strassen"), (2, "slabs"), (2, "auto"), (4, "slabs"), (3, "slabs")])
def test_bench_ranks_on_one_gpu(world, variant):
    """bench.py's N > 1 path, ranks as processes sharing GPU 0, transport = gloo staged through the host
    (RCCL refuses two ranks on one device); --check compares every rank's part of C with the product the rank
    recomputes alone."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # launched the way the driver launches it: a torch.distributed.run around the command (it picks its own rendezvous port).  Launcher
    # rank 0 becomes the controller of the run, starts the ranks that do the work and prints their line; the other launcher ranks step aside
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={world}",
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--size", "8192", "--backend", "gloo", "--check", "--variant", variant, "--no-cpu-baseline", "--no-links", "--no-n1", "--transport", "rccl"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("-> OK") == world and r.stdout.count('"metric"') == 1, r.stdout[-3000:]


@pytest.mark.parametrize("m,l,n,seeds,world", [(65536, 65536, 65536, (3, 4), 8),        # BASELINE.json configs[3]: the sharded execution path
                                               (131072, 8192, 131072, (5, 6), 8)])      # BASELINE.json configs[4]
def test_baseline_sizes_through_the_sharded_path_vs_reference_sha256(m, l, n, seeds, world):
    """The two 8-GPU configurations of BASELINE.json at full size through the multi-device path (8 ranks on this one
    GPU: own streams, slabs, peer copies, sub-products), against the SHA-256 of the real reference's product
    (tests/golden/sha256.json)."""
    import hashlib
    import json
    path = os.path.join(ROOT, "tests", "golden", "sha256.json")
    want = [e["sha256"] for e in json.load(open(path)) if (e["op"], e["m"], e["l"], e["n"], e["seed_a"], e["seed_b"]) == ("mul", m, l, n, *seeds)]
    assert want, "no golden SHA-256 for this product"
    m4ri_amd.set_devices([0] * world)
    A, B = Mzd.random(m, l, seeds[0]), Mzd.random(l, n, seeds[1])
    C = Mzd.init(m, n)
    m4ri_amd.mul_multi(C, A, B, False, 0, 0)
    assert hashlib.sha256(C.masked().tobytes()).hexdigest() == want[0]
    # the schedule is chosen behind the C boundary: configs[3] takes the Strassen sub-products, configs[4] (l = 8192) row slabs
    st = m4ri_amd.multi_stats()
    assert st.variant == (m4ri_amd.VARIANT_SLABS if l == 8192 else m4ri_amd.VARIANT_STRASSEN) and st.world == world and st.converted == 0


# ---- bench.py as the driver runs it: the command itself starts the ranks ---------------------------------------------
def _bench(args, timeout=900, lean=True):
    """`python bench.py ...` exactly as typed (no launcher around it); returns rank 0's JSON line and the whole stdout.  lean: without the
    controller's link probe and one-GPU run (two more processes per call; the tests that are about them pass lean=False)."""
    import json
    if lean:
        args = list(args) + [f for f in ("--no-links", "--no-n1") if f not in args]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0]), r.stdout


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it runs two ranks and says so (the reference switches to its
    multi-core path inside the same command, bench/bench_multiplication.c:94-103); a --gpus request is never answered with
    a one-GPU line."""
    out, stdout = _bench(["--gpus", "2", "--size", "8192", "--backend", "gloo", "--check", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--virtual-ranks"], lean=False)
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["variant"] == "slabs"
    assert "all_gather" in out["config"]["collective"] and stdout.count("-> OK") == 2
    assert out["config"]["transport"] == "rccl" and out["config"]["transport_fallback"] == [] and out["host_issue_ms_per_step"] > 0
    # the headline is one product at a time; the stream of products (two in flight) is a second, separately named number on every N > 1 line
    assert out["config"]["inflight"] == 1 and out["pipelined_value"] > 0 and out["pipelined_ms_per_step"] > 0
    assert out["speedup_vs_n1"]["n1_ms_per_step"] > 0 and out["config"]["links"]["ranks_share_devices"] is True and out["config"]["controller_wall_s"] > 0
    out, stdout = _bench(["--gpus", "2", "--size", "8192", "--backend", "gloo", "--check", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                          "--inflight", "1", "--no-links", "--no-n1"])
    assert "pipelined_value" not in out and "links" not in out["config"] and "speedup_vs_n1" not in out and stdout.count("-> OK") == 2
    out, stdout = _bench(["--gpus", "2", "--size", "8192", "--backend", "gloo", "--check", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--variant", "strassen", "--overlap", "2"])
    assert out["n_gpus"] == 2 and out["config"]["overlap_chunks"] == [2, 1] and stdout.count("-> OK") == 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--size", "8192", "--no-cpu-baseline", "--inner"], capture_output=True,
                       text=True, env=dict(os.environ, WORLD_SIZE="1", RANK="0"), timeout=300, cwd=ROOT)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout   # a rank started alone for --gpus 8: refuse, do not print n_gpus 1


def test_bench_overlapped_strassen_schedule_at_8_ranks():
    """BASELINE.json configs[3]'s execution path at 8 ranks with the transport overlapped (two row chunks per sub-product:
    2 row x 2 column units per sub-product: the operand chunks of later units and the results of earlier ones travel under the multiplications): every rank's slabs of C against the
    product it recomputes alone.  (gloo on one GPU completes every batch when it is posted: the bits and the batch order are
    what is tested here, the overlap itself needs links.)"""
    out, stdout = _bench(["--gpus", "8", "--size", "16384", "--backend", "gloo", "--check", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--overlap", "2x2", "--inflight", "2", "--shard-levels", "1"], timeout=1500)
    assert out["n_gpus"] == 8 and out["config"]["variant"] == "strassen" and out["config"]["overlap_chunks"] == [2, 2] and out["config"]["inflight"] == 1
    assert out["config"]["sub_products"] == 7 and stdout.count("-> OK") == 8


def test_bench_config4_at_8_ranks_full_size_matches_the_reference():
    """BASELINE.json configs[3] exactly as the driver's 8-GPU command runs it -- 65536^3, `--variant auto` = the 47 sub-products of the
    top TWO levels done as one application of the rank-47 scheme (6 rounds on 8 ranks; a rank's sub-products of 16384^3 in batched
    products), slab-cyclic layout -- minus the links (8 processes on this one GPU, gloo): the C gathered from the 8 ranks against the
    real reference's SHA-256, and every rank's slabs against the product it recomputes alone."""
    out, stdout = _bench(["--gpus", "8", "--backend", "gloo", "--check", "--steps", "1", "--warmup", "1"], timeout=2400)
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["variant"] == "strassen" and cfg["sub_products"] == 47 and cfg["overlap_chunks"] == [1, 1]
    assert cfg["sharded_levels"] == 2 and cfg["sub_products_on_busiest_rank"] == 6 and cfg["sub_products_per_batched_product"] >= 2
    assert cfg["per_rank_product"] == [16384, 16384, 16384] and cfg["bytes_over_links_per_step"] == 3 * 47 * 7 * (4 << 20)
    assert out["verified"]["matches_reference"] is True and stdout.count("-> OK") == 8


def test_bench_config4_one_sharded_level_at_8_ranks_full_size_matches_the_reference():
    """The same with ONE sharded level on request (`--shard-levels 1`): the 7 sub-products of the top Strassen-Winograd level, one per
    rank, two row chunks per sub-product in flight."""
    out, stdout = _bench(["--gpus", "8", "--backend", "gloo", "--check", "--steps", "1", "--warmup", "1", "--shard-levels", "1", "--no-cpu-baseline"], timeout=2400)
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["variant"] == "strassen" and cfg["sub_products"] == 7 and cfg["overlap_chunks"] == [2, 1]
    assert cfg["per_rank_product"] == [32768, 32768, 32768] and cfg["bytes_over_links_per_step"] == 3 * 7 * 7 * (16 << 20)
    assert out["verified"]["matches_reference"] is True and stdout.count("-> OK") == 8


def test_bench_config5_at_8_ranks_takes_row_slabs_and_matches_the_reference():
    """BASELINE.json configs[4] (131072 x 8192 x 131072) at 8 ranks: `auto` = row slabs of A and C + ONE all-gather of B
    (the reference's row parallelism, m4ri/brilliantrussian.c:1121-1123), never the Strassen split; the gathered
    C against the real reference's SHA-256, and every rank's slab against the product it recomputes alone."""
    out, stdout = _bench(["--workload", "rect131072", "--gpus", "8", "--backend", "gloo", "--check", "--steps", "1", "--warmup", "1"], timeout=1500)
    assert out["n_gpus"] == 8 and out["config"]["variant"] == "slabs" and out["config"]["per_rank_product"] == [16384, 8192, 131072]
    assert out["verified"]["matches_reference"] is True and stdout.count("-> OK") == 8


def test_bench_ragged_slabs_match_the_reference():
    """Row slabs when the world size divides nothing: 100003 x 50021 x 70017 over 3 ranks (slabs of 33335 / 33335 / 33333 rows
    of A, 16674 / 16674 / 16673 of B, the gathered B padded), gathered C vs the reference's SHA-256."""
    out, stdout = _bench(["--dims", "100003,50021,70017", "--seeds", "21,22", "--gpus", "3", "--backend", "gloo", "--check", "--steps", "1",
                          "--warmup", "1", "--no-cpu-baseline"], timeout=1500)
    assert out["n_gpus"] == 3 and out["config"]["variant"] == "slabs" and out["config"]["slab_rows"] == [33335, 16674]
    assert out["verified"]["matches_reference"] is True and stdout.count("-> OK") == 3


# ---- RCCL on the lease: the nccl backend at world size 1 -------------------------------------------------------------------
_NCCL_PROBE = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from m4ri_amd import sharding
x = sharding.torch_exchange(dist)
x([], [])                                                   # an empty batch is a no-op, not an RCCL call
a = torch.arange(1 << 16, dtype=torch.int64, device="cuda"); b = torch.zeros_like(a)
x([(0, a)], [(0, b)])                                       # one self-addressed send/recv pair in one group
torch.cuda.synchronize(); assert torch.equal(a, b)
c = torch.zeros_like(a); d = torch.zeros((64, 1024), dtype=torch.int64, device="cuda")
h1 = x.post([(0, a), (0, a[:4096])], [(0, c), (0, d[:, :64])])    # two pairs, one into a NON-contiguous view (temporary + copy back)
h2 = x.post([(0, c)], [(0, b)])                             # a second batch queued behind the first before either is waited for
h1.wait(); h2.wait(); torch.cuda.synchronize()
assert torch.equal(c, a) and torch.equal(d[:, :64].reshape(-1), a[:4096]) and not d[:, 64:].any()
mine = torch.arange(5 * 7, dtype=torch.int64, device="cuda").reshape(5, 7); full = torch.empty((5, 7), dtype=torch.int64, device="cuda")
sharding.all_gather_rows(dist, full, mine)                  # the slabs variant's one collective
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks
dist.barrier(); torch.cuda.synchronize()
assert torch.equal(full, mine) and float(t.item()) == 1.5
dist.destroy_process_group()
print("RCCL world-size-1 probe OK")
'''


def test_rccl_backend_at_world_size_1():
    """The transport code of the N > 1 path (sharding.torch_exchange incl. its asynchronous post/wait, all_gather_rows, the
    barrier and the max-reduce of bench.py) through the real nccl (= RCCL) backend, the only way a one-GPU box can: world
    size 1, self-addressed batches.  API misuse shows up here, not in the driver's 8-GPU run."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    r = subprocess.run([sys.executable, "-c", _NCCL_PROBE, ROOT, str(port)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "probe OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("variant", ["strassen", "slabs"])
def test_bench_distributed_code_path_under_rccl_at_world_size_1(variant):
    """bench.py's whole N > 1 code path (process group, fences, batches, collective, timing reduce, gathered-C check) with the
    nccl backend at world size 1 (--force-dist): what the driver's multi-GPU run executes, minus the links."""
    out, stdout = _bench(["--gpus", "1", "--force-dist", "--variant", variant, "--size", "16384", "--steps", "2", "--warmup", "1", "--check"])
    assert out["n_gpus"] == 1 and out["config"]["variant"] == variant and "RCCL" in out["config"]["backend"] and stdout.count("-> OK") == 1


# ---- the transport ladder of the N > 1 command ---------------------------------------------------------------------------
def test_bench_peer_transport_one_process_all_ranks():
    """`bench.py --gpus 8 --transport peer`: ONE process drives all ranks through m4ri_amd_dmat_mul (the schedules behind the C
    boundary; here 8 virtual ranks on this GPU), prints the same line -- schedule chosen in C, per-rank timeline, host issue time."""
    out, stdout = _bench(["--gpus", "8", "--transport", "peer", "--virtual-ranks", "--size", "16384", "--check", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"])
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["transport"] == "peer" and cfg["variant"] == "strassen" and cfg["transport_fallback"] == []
    assert cfg["schedule_stats"]["variant"] == "strassen" and cfg["schedule_stats"]["sub_products"] == 47 and cfg["schedule_stats"]["operands_converted"] == 0
    assert cfg["schedule_stats"]["sharded_levels"] == 2 and cfg["schedule_stats"]["sub_products_per_batched_product"] >= 1
    assert len(cfg["timeline_ms_last_lane0_product"]) == 8 and out["host_issue_ms_per_step"] > 0 and stdout.count("-> OK") == 1
    assert out["pipelined_value"] > 0 and cfg["schedule_stats"]["rank_pairs_copying_through_the_host"] == 0   # two lanes, both Cs checked by --check
    out, stdout = _bench(["--gpus", "4", "--transport", "peer", "--virtual-ranks", "--dims", "20000,8192,30016", "--check", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-links", "--no-n1"])
    assert out["n_gpus"] == 4 and out["config"]["schedule_stats"]["variant"] == "slabs" and stdout.count("-> OK") == 1


def test_bench_8_gpu_line_is_complete_before_hardware_sees_it():
    """`bench.py --gpus 8` exactly as the driver types it (auto transport, real backend: peer first), here on 8 virtual
    ranks of one GPU at a reduced size -- the ONE line carries the reference's CPU baseline, the roofline objects, the measured links,
    one product AND the stream of products, the speed-up over the same binary on one GPU, and the controller's wall time; the second rung
    (RCCL needs one device per rank) is noted, not fatal."""
    out, stdout = _bench(["--gpus", "8", "--virtual-ranks", "--size", "16384", "--steps", "3", "--warmup", "1", "--watchdog", "90"], timeout=1500, lean=False)
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["transport"] == "peer" and cfg["transport_fallback"] == [] and cfg["variant"] == "strassen"
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["kind"] in ("reference", "port")
    assert out["roofline"]["frac"] > 0 and out["roofline"]["launch_ms"] > 0 and out["roofline_schedule"]["frac"] > 0
    assert out["roofline_schedule"]["north_star_60pct"] is False and "frac_of_copy_peak" not in out["roofline_schedule"]
    links = cfg["links"]
    assert links["pairs"] == 56 and links["gbs_per_direction_min"] > 0 and links["all_at_once_gbs"] > 0 and len(links["peer_access"]) == 8
    assert links["ranks_share_devices"] is True and "blit" in links["what"]
    assert out["pipelined_value"] > 0 and out["value"] > 0 and out["speedup_vs_n1"]["n1_ms_per_step"] > 0 and out["speedup_vs_n1"]["one_product"] > 0
    assert 0 < cfg["controller_wall_s"] < 600
    assert cfg.get("transports_unavailable", [{}])[0].get("transport") == "rccl" or "rccl" in cfg.get("transports_measured", {})


@pytest.mark.parametrize("inject", ["crash", "hang"])
def test_bench_falls_back_down_the_ladder_and_still_prints_one_line(inject):
    """The first rung (one process per rank over torch.distributed) fails -- its ranks die, or never finish and the watchdog kills them
    (M4RI_AMD_BENCH_INJECT, a test hook in the ranks) -- and the command still ends with exactly one result line, from the peer
    transport, saying what happened; with no rung left it ends with one error line and a non-zero exit code."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", M4RI_AMD_BENCH_INJECT=inject)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "8192", "--backend", "gloo", "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--watchdog", "90" if inject == "crash" else "45"]
    r = subprocess.run(base + ["--virtual-ranks", "--check"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["transport"] == "peer" and out["config"]["transport_fallback"][0]["transport"] == "rccl"
    assert ("watchdog" in out["config"]["transport_fallback"][0]["reason"]) == (inject == "hang")
    if inject == "crash":   # nothing left to fall back to: one error line, non-zero exit code
        r = subprocess.run(base + ["--transport", "rccl"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode != 0 and len(lines) == 1 and "error" in json.loads(lines[0]) and '"metric"' not in r.stdout
