import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, m4ri_amd
m4ri_amd.init(0)
for n in (2048, 4096, 8192, 12288, 16384, 24576, 32768):
    w = n // 64
    A = torch.empty((n, w), dtype=torch.int64, device="cuda"); B = torch.empty_like(A); C = torch.empty_like(A)
    m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 3); m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 4)
    for _ in range(3): m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
    torch.cuda.synchronize(); reps = 20
    t = time.perf_counter()
    for _ in range(reps): m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print(f"{n:6d}: {dt*1e3:8.3f} ms  {n**3/dt:.3e}  fp={int(C.sum().item()) & 0xffffffff:08x}", flush=True)
