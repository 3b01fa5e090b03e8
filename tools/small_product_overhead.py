import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, m4ri_amd
m4ri_amd.init(0)
for n in (512, 2048, 4096):
    w = n // 64
    A = torch.empty((n, w), dtype=torch.int64, device="cuda"); B = torch.empty((n, w), dtype=torch.int64, device="cuda"); C = torch.empty((n, w), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 3); m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 4)
    for _ in range(6):
        m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50):
        m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(n, "host issue per call %.1f us, total per call %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6), flush=True)
