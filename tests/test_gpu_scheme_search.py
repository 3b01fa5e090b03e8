"""tools/flipgraph_444_gpu.hip -- the flip-graph walks that look for the table of scheme_passes.hip, one wavefront per walk.
A few seconds of it from the standard algorithm must come down to rank <= 52, and what it prints must be a scheme of the 4 x 4 x 4 product
over GF(2) by the definition (the tool verifies every scheme on the host itself; this is the second opinion, in numpy)."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_walks_on_the_gpu_come_down_from_the_standard_algorithm_and_print_schemes():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(ROOT, "build", "flipgraph_444_gpu_test")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "flipgraph_444_gpu.hip"), "-o", exe], check=True, timeout=300)
    # 6 s, no pool files, path limit 5e6, no plus transitions, 4096 walks, 50 000 flips per launch, from the standard algorithm, span 3
    out = subprocess.run([exe, "6", "none", "none", "5000000", "0", "0", "4096", "50000", "x", "3"], capture_output=True, text=True, timeout=120)
    assert out.returncode in (0, 1), out.stderr[-2000:]   # 0: rank 47 reached (!), 1: not; 4 = a scheme from the device did not verify
    blocks = re.split(r"^# rank (\d+) after.*$", out.stdout, flags=re.M)
    ranks = [int(r) for r in blocks[1::2]]
    assert ranks and ranks == sorted(ranks, reverse=True) and ranks[-1] <= 52, ranks
    tri = re.findall(r"\{0x([0-9a-f]{4}), 0x([0-9a-f]{4}), 0x([0-9a-f]{4})\},", blocks[-1])
    assert len(tri) == ranks[-1]
    _check_scheme(tri, 4)


@pytest.mark.gpu
def test_walks_on_the_gpu_find_rank_23_for_3x3x3():
    """The check of the moves themselves: over GF(2) the 3 x 3 x 3 product comes down from 27 to 23 multiplications by flips and reductions in
    well under a second when EVERY group a flip changes is tested for linear dependence (five of them; with three the walk stops at 26)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(ROOT, "build", "flipgraph_333_gpu_test")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-DNDIM=3", os.path.join(ROOT, "tools", "flipgraph_444_gpu.hip"), "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe, "20", "none", "none", "2000000", "0", "0", "4096", "50000", "x", "3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])   # 0 = the target (23) reached
    blocks = re.split(r"^# rank (\d+) after.*$", out.stdout, flags=re.M)
    assert int(blocks[-2]) == 23
    _check_scheme(re.findall(r"\{0x([0-9a-f]{4}), 0x([0-9a-f]{4}), 0x([0-9a-f]{4})\},", blocks[-1]), 3)


def _check_scheme(tri, n):
    U, V, W = (np.array([int(t[f], 16) for t in tri], dtype=np.uint32) for f in range(3))
    bit = lambda a, k: ((a >> np.uint32(k)) & np.uint32(1)).astype(np.uint8)   # noqa: E731
    for i in range(n):
        for j in range(n):
            for j2 in range(n):
                for k in range(n):
                    for i2 in range(n):
                        for k2 in range(n):
                            got = int((bit(U, n * i + j) & bit(V, n * j2 + k) & bit(W, n * i2 + k2)).sum() & 1)
                            assert got == int(i == i2 and j == j2 and k == k2), (i, j, j2, k, i2, k2)


def test_the_host_walks_find_rank_23_for_3x3x3():
    """The same check of the moves for the host tool (tools/flipgraph_444.c, -DN=3), no GPU needed."""
    exe = os.path.join(ROOT, "build", "flipgraph_333_test")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-pthread", "-DN=3", os.path.join(ROOT, "tools", "flipgraph_444.c"), "-o", exe], check=True, timeout=120)
    # 2 threads, at most 60 s, target 23, from the standard algorithm, no checkpoints, path limit 1e6, no plus transitions, general reduction
    out = subprocess.run([exe, "2", "60", "23", "x", "none", "none", "1000000", "0", "1", "3", "1", "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-1500:]
    blocks = re.split(r"^# rank (\d+) after.*$", out.stdout, flags=re.M)
    assert int(blocks[-2]) == 23
    _check_scheme(re.findall(r"\{0x([0-9a-f]{4}), 0x([0-9a-f]{4}), 0x([0-9a-f]{4})\},", blocks[-1]), 3)
