import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, m4ri_amd
m4ri_amd.init(0)
for n in (16384, 32768, 65536):
    w = n // 64
    A = torch.empty((n, w), dtype=torch.int64, device="cuda"); B = torch.empty_like(A); C = torch.zeros_like(A)
    m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 3); m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 4)
    for add in (False, True):
        for _ in range(2): m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, add=add)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, add=add)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(n, "addmul" if add else "mul   ", f"{dt*1e3:.3f} ms", flush=True)
