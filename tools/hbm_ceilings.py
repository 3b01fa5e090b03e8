#!/usr/bin/env python3
"""What this box's HBM gives a streaming kernel, by direction: write-only (memset), read-only (a reduction), copy -- the
ceilings the Winograd passes are priced against (the down passes write 343/64 = 5.4 words per word they read, the up
pass reads 5.4 per word written).  torch kernels only; GB/s counts read + written bytes."""
import torch

dev = "cuda"
n = 1 << 29  # int64 words: 4 GiB
x = torch.empty(n, dtype=torch.int64, device=dev)
y = torch.empty(n, dtype=torch.int64, device=dev)


def best_ms(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    b = 1e9
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        b = min(b, e0.elapsed_time(e1))
    return b


for name, fn, nbytes in (("write only  (x.zero_())", lambda: x.zero_(), 8 * n),
                         ("write only  (x.fill_(7))", lambda: x.fill_(7), 8 * n),
                         ("read only   (x.sum())", lambda: x.sum(), 8 * n),
                         ("copy        (y.copy_(x))", lambda: y.copy_(x), 16 * n),
                         ("xor         (y ^= x: 2 reads + 1 write)", lambda: y.bitwise_xor_(x), 24 * n)):
    ms = best_ms(fn)
    print(f"{name:44s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
