"""Quick device-resident timing of m4ri_amd_transpose_dev (one size: python tools/transpose_quick.py 65536)."""
import sys, torch, m4ri_amd
m4ri_amd.init(0); torch.cuda.set_device(0); L = m4ri_amd.lib()
def ev(fn, reps=20 if len(sys.argv) == 1 else 3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
import os
PAD_A, PAD_D = int(os.environ.get('PAD_A', 0)), int(os.environ.get('PAD_D', 0))
shapes = ((16384, 16384), (65536, 32768), (65536, 65536), (100000, 50000))
if len(sys.argv) > 1:
    shapes = ((int(sys.argv[1]), int(sys.argv[1])),)
for m, n in shapes:
    sa, sd = (n + 63) // 64, (m + 63) // 64
    sa += sa & 1; sd += sd & 1
    sa += PAD_A; sd += PAD_D
    A = torch.randint(-2**62, 2**62, (m, sa), dtype=torch.int64, device="cuda")
    D = torch.empty((n, sd), dtype=torch.int64, device="cuda")
    ms = ev(lambda: L.m4ri_amd_transpose_dev(D.data_ptr(), sd, A.data_ptr(), sa, m, n, None))
    byt = 8.0 * (m * ((n + 63) // 64) + n * ((m + 63) // 64))
    print(f"  {m:6d} x {n:6d} (stride pads {PAD_A}, {PAD_D}): {ms:8.3f} ms  {byt / ms / 1e9:7.3f} TB/s")
