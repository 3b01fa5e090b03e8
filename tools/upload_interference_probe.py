#!/usr/bin/env python3
"""Developer probe: does an H2D upload running beside it slow a resident product down, and does it matter whether the source is pageable or
pinned?  (The block pipeline of mzd_mul from host memory sees its first products, which run beside the uploads, ~8 % slower than the last.)
Resident 32768^3 products timed by events on their stream while a second thread copies 128 MiB blocks host -> device in a loop."""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import m4ri_amd
m4ri_amd.init(0)
n = 32768
w = n // 64
A = torch.empty((n, w), dtype=torch.int64, device="cuda"); B = torch.empty((n, w), dtype=torch.int64, device="cuda"); C = torch.empty((n, w), dtype=torch.int64, device="cuda")
m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 3); m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 4)
st = torch.cuda.Stream()
def products(k):
    out = []
    with torch.cuda.stream(st):
        for _ in range(k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, False, 0, st.cuda_stream)
            e1.record(st)
            out.append((e0, e1))
    st.synchronize()
    return [a.elapsed_time(b) for a, b in out]
products(4)
print("alone:                 " + " ".join(f"{x:.3f}" for x in products(8)))
words = (128 << 20) // 8
dst = torch.empty(words, dtype=torch.int64, device="cuda")
for kind in ("pageable", "pinned", "d2h pageable"):
    src = torch.ones(words, dtype=torch.int64)
    if kind == "pinned":
        src = src.pin_memory()
    stop = False
    moved = [0]
    def pump():
        cs = torch.cuda.Stream()
        with torch.cuda.stream(cs):
            while not stop:
                if kind.startswith("d2h"):
                    src.copy_(dst, non_blocking=False)
                else:
                    dst.copy_(src, non_blocking=False)
                cs.synchronize()
                moved[0] += 1
    t = threading.Thread(target=pump); t.start()
    time.sleep(0.05)
    t0 = time.perf_counter(); m0 = moved[0]
    ts = products(8)
    dt = time.perf_counter() - t0; m1 = moved[0]
    stop = True; t.join()
    print(f"beside {kind:13s} copies: " + " ".join(f"{x:.3f}" for x in ts) + f"   ({(m1 - m0) * 0.125 / dt:.1f} GiB/s moved meanwhile)")
