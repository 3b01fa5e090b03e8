#!/usr/bin/env python3
"""HBM rate of the streaming elimination kernels on device-resident data:
  * m4ri_amd_process_rows_dev (the twin of mzd_process_rows6: six 8-bit tables, k = 48) over an n x n matrix,
  * the PLE's rank-64 update, through one 64-column block step of m4ri_amd_ple_dev on a matrix whose first block is
    full rank (timed as a whole PLE of n x 128: one update of n x 64 ... use rocprofv3 for the kernel alone).
Algorithmic bytes of a row-processing pass = 2 * 8 * rows * words (every touched word read once and written once).
usage: elim_bench.py [n]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import m4ri_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    torch.cuda.set_device(0)
    m4ri_amd.init(0)
    L = m4ri_amd.lib()
    st = torch.cuda.current_stream().cuda_stream
    w = n // 64
    M = torch.empty((n, w), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(M.data_ptr(), w, n, n, 1, st)
    for nt in (6, 2, 1):
        kb = [8] * nt
        T = [torch.empty((256, w), dtype=torch.int64, device="cuda") for _ in range(nt)]
        for i, t in enumerate(T):
            m4ri_amd.fill_dev(t.data_ptr(), w, 256, n, 10 + i, st)
        Ls = [torch.from_numpy(np.random.default_rng(i).permutation(256).astype(np.int32)).cuda() for i in range(nt)]
        idx = torch.empty(6 * n, dtype=torch.int32, device="cuda")
        kbits = (ctypes.c_int32 * 6)(*(kb + [0] * (6 - nt)))
        tp = (ctypes.c_void_p * 6)(*([t.data_ptr() for t in T] + [0] * (6 - nt)))
        ts = (ctypes.c_int64 * 6)(*([w] * nt + [0] * (6 - nt)))
        lp = (ctypes.c_void_p * 6)(*([l.data_ptr() for l in Ls] + [0] * (6 - nt)))
        for startcol in (0, n // 2):
            def run():
                rc = L.m4ri_amd_process_rows_dev(M.data_ptr(), w, w, 0, n, startcol, nt, kbits, tp, ts, lp, idx.data_ptr(), st)
                assert rc == 0, rc
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                e0.record(); run(); e1.record(); e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            words = w - startcol // 64
            gb = 2 * 8 * n * words / 1e9
            print(f"process_rows{nt if nt > 1 else ''} n={n} startcol={startcol}: {best:.3f} ms, {gb / best:.2f} TB/s algorithmic "
                  f"({gb / best / 8 * 100:.0f} % of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
