#!/usr/bin/env python3
"""PCIe copy rates on the box: pageable vs pinned, one direction vs both at once (two host threads)."""
import threading
import time

import torch

N = 512 << 20
dev = torch.empty(N, dtype=torch.uint8, device="cuda")
dev2 = torch.empty(N, dtype=torch.uint8, device="cuda")
page = torch.empty(N, dtype=torch.uint8); page.fill_(1)
page2 = torch.empty(N, dtype=torch.uint8); page2.fill_(2)
pin = torch.empty(N, dtype=torch.uint8).pin_memory()
pin2 = torch.empty(N, dtype=torch.uint8).pin_memory()


def rate(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return N / best / 2 ** 30


print(f"H2D pageable {rate(lambda: dev.copy_(page)):.1f} GiB/s   pinned {rate(lambda: dev.copy_(pin, non_blocking=True)):.1f} GiB/s")
print(f"D2H pageable {rate(lambda: page.copy_(dev)):.1f} GiB/s   pinned {rate(lambda: pin.copy_(dev, non_blocking=True)):.1f} GiB/s")


def both(h2d_src, d2h_dst, nb):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def up():
        with torch.cuda.stream(s1):
            dev.copy_(h2d_src, non_blocking=nb); s1.synchronize()
    def down():
        with torch.cuda.stream(s2):
            d2h_dst.copy_(dev2, non_blocking=nb); s2.synchronize()
    def run():
        a, b = threading.Thread(target=up), threading.Thread(target=down)
        a.start(); b.start(); a.join(); b.join()
    return run


print(f"both directions at once, pageable: {2 * rate(both(page, page2, False)):.1f} GiB/s aggregate;  pinned: {2 * rate(both(pin, pin2, True)):.1f} GiB/s aggregate")
# host memcpy rate (one thread) into pinned memory: the staging step a do-it-yourself pipeline would add
t = time.perf_counter(); pin.copy_(page); dt = time.perf_counter() - t
print(f"host memcpy pageable -> pinned, one thread: {N / dt / 2 ** 30:.1f} GiB/s")
