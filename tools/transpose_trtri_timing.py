"""Device-resident and host-API timings of the transpose and of the triangular inverse.
transpose: algorithmic bytes = 8 * (m * W(n) + n * W(m)) per launch, against HBM (8 TB/s vendor).
usage: python tools/transpose_trtri_timing.py [--quick]"""
import sys
import time

import numpy as np
import torch

import m4ri_amd
from m4ri_amd.mzd import Mzd


def ev_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    quick = "--quick" in sys.argv
    m4ri_amd.init(0)
    torch.cuda.set_device(0)
    L = m4ri_amd.lib()
    print("== m4ri_amd_transpose_dev, device resident (events on the null stream)")
    for m, n in ((4096, 4096), (16384, 16384), (65536, 32768), (32768, 65536), (65536, 65536), (65536, 65472), (100000, 50000)):
        sa, sd = (n + 63) // 64, (m + 63) // 64
        sa += sa & 1
        sd += sd & 1
        A = torch.randint(-2**62, 2**62, (m, sa), dtype=torch.int64, device="cuda")
        D = torch.empty((n, sd), dtype=torch.int64, device="cuda")
        ms = ev_time(lambda: L.m4ri_amd_transpose_dev(D.data_ptr(), sd, A.data_ptr(), sa, m, n, None), 20)
        byt = 8.0 * (m * ((n + 63) // 64) + n * ((m + 63) // 64))
        # the plain copy of the same bytes on the same box, for scale
        C = torch.empty_like(A)
        ms_copy = ev_time(lambda: C.copy_(A), 20)
        print(f"  {m:6d} x {n:6d}: {ms:8.3f} ms  {byt / ms / 1e9:7.3f} TB/s ({byt / ms / 1e9 / 8.0:.3f} of 8 TB/s); torch copy of A: {2 * A.numel() * 8 / ms_copy / 1e9:.3f} TB/s")
        del A, D, C
    print("== m4ri_amd_trtri_upper_dev, device resident")
    for n in ((4096, 16384) if quick else (4096, 16384, 32768, 65536)):
        s = n // 64
        U = torch.randint(-2**62, 2**62, (n, s), dtype=torch.int64, device="cuda")
        W = U.clone()

        def run():
            W.copy_(U)
            L.m4ri_amd_trtri_upper_dev(W.data_ptr(), s, n, None)
        ms_both = ev_time(run, 5)
        ms_copy = ev_time(lambda: W.copy_(U), 5)
        print(f"  n = {n:6d}: {ms_both - ms_copy:8.3f} ms")
        del U, W
    print("== host API (host memory in, host memory out; best of 3; the transpose into an existing, touched DST)")
    for n in ((4096, 16384) if quick else (4096, 16384, 32768, 65536)):
        A, DST = Mzd.random(n, n, 1), Mzd.random(n, n, 3)
        t1 = []
        for _ in range(4):
            t = time.perf_counter(); m4ri_amd.mzd_transpose(A, DST); t1.append(time.perf_counter() - t)
        U = Mzd.random(n, n, 2)
        idx = np.arange(n)
        U.valid_words()[idx, idx // 64] |= np.uint64(1) << (idx % 64).astype(np.uint64)
        t2 = []
        for _ in range(4):
            V = U.copy()
            t = time.perf_counter(); m4ri_amd.mzd_trtri_upper(V); t2.append(time.perf_counter() - t)
        print(f"  n = {n:6d}: mzd_transpose {min(t1[1:]) * 1e3:9.2f} ms   mzd_trtri_upper {min(t2[1:]) * 1e3:9.2f} ms   (all: {[round(x * 1e3, 1) for x in t1]} {[round(x * 1e3, 1) for x in t2]})")


if __name__ == "__main__":
    main()
