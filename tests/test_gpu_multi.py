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


@pytest.mark.parametrize("world,variant,layout", [(2, "strassen", "distributed"), (2, "strassen", "owner"), (2, "blocks", "owner"),
                                                  (4, "strassen", "distributed"), (2, "slabs", "distributed"), (4, "slabs", "owner"),
                                                  (2, "auto", "distributed")])
def test_bench_ranks_on_one_gpu(world, variant, layout):
    """bench.py's N > 1 path, ranks as processes sharing GPU 0, transport = gloo staged through the host
    (RCCL refuses two ranks on one device); --check compares every rank's part of C with the product the rank
    recomputes alone."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--size", "8192", "--backend", "gloo", "--check", "--variant", variant, "--layout", layout, "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("-> OK") == world, r.stdout[-3000:]


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
