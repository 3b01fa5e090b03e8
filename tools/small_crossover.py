import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import m4ri_amd
from m4ri_amd.mzd import Mzd
def best(fn, reps):
    fn(); b = 1e30
    for _ in range(reps):
        t = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t)
    return b
m4ri_amd.init(0)
shapes = [(n, n, n) for n in (32, 64, 128, 192, 256, 320, 384, 448, 512, 576, 640, 768, 1024)] + [
    (1024, 256, 256), (256, 1024, 256), (256, 256, 1024), (2048, 2048, 16), (16, 2048, 2048), (2048, 16, 2048), (1000, 10, 20), (16, 4096, 16),
    (4096, 16, 64), (64, 64, 4096), (2048, 64, 64), (512, 512, 8), (100, 1000, 100), (200, 200, 1000), (1024, 64, 1024), (64, 16384, 64), (4096, 64, 4096), (128,128,8192), (8192,128,128)]
print("shape | gpu path us | host routine us (both through mzd_mul from python)")
for (m, l, n) in shapes:
    A, B, C = Mzd.random(m, l, 3), Mzd.random(l, n, 4), Mzd.init(m, n)
    m4ri_amd.set_small_product_threshold(0)
    tg = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 50)
    want = C.copy()
    m4ri_amd.set_small_product_threshold(1 << 62)
    th = best(lambda: m4ri_amd.mzd_mul(C, A, B, 0), 50)
    assert C.equal(want)
    print(f"{m}x{l}x{n} | {tg*1e6:8.1f} | {th*1e6:8.1f}", flush=True)
