import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, m4ri_amd
m4ri_amd.init(0)
tag = os.environ.get("M4RI_AMD_SMALL_LEAF", "auto")
for (m, l, n) in [(64, 1 << 20, 64), (1, 1, 1 << 26), (1 << 20, 8, 8), (100000, 1, 600), (1, 1 << 20, 1), (3, 100000, 5000), (5000, 100000, 3), (16, 16, 1 << 20), (1 << 18, 64, 64), (1000, 16000, 1000), (130, 130000, 130), (257, 513, 120000)]:
    wl, wn = (l + 63) // 64, (n + 63) // 64
    A = torch.empty((m, wl), dtype=torch.int64, device="cuda"); B = torch.empty((l, wn), dtype=torch.int64, device="cuda"); C = torch.zeros((m, wn), dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(A.data_ptr(), wl, m, l, 3); m4ri_amd.fill_dev(B.data_ptr(), wn, l, n, 4)
    out = []
    for add in (False, True):
        for _ in range(5): m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, add, 0)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 20 * 1e6)
    C.zero_(); m4ri_amd.mul_dev(C.data_ptr(), wn, A.data_ptr(), wl, B.data_ptr(), wn, m, l, n, False, 0); torch.cuda.synchronize()
    w = torch.arange(1, C.numel() + 1, dtype=torch.int64, device="cuda").reshape(C.shape)
    print(f"{tag} {m}x{l}x{n}: mul {out[0]:8.1f} us addmul {out[1]:8.1f} us gen {m4ri_amd.get_stats().leaf_gen} sum {int((C * w).sum().item()) & 0xffffffffffff:012x}", flush=True)
