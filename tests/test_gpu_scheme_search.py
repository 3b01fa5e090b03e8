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
    U, V, W = (np.array([int(t[f], 16) for t in tri], dtype=np.uint32) for f in range(3))
    bit = lambda a, k: ((a >> np.uint32(k)) & np.uint32(1)).astype(np.uint8)   # noqa: E731
    for i in range(4):
        for j in range(4):
            for j2 in range(4):
                for k in range(4):
                    for i2 in range(4):
                        for k2 in range(4):
                            got = int((bit(U, 4 * i + j) & bit(V, 4 * j2 + k) & bit(W, 4 * i2 + k2)).sum() & 1)
                            assert got == int(i == i2 and j == j2 and k == k2), (i, j, j2, k, i2, k2)
