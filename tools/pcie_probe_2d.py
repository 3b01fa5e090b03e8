#!/usr/bin/env python3
"""hipMemcpy vs hipMemcpy2D from/to pageable host memory (what the host API of the library issues)."""
import ctypes
import time

import numpy as np
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy2D.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
rows, pitch = 65536, 8192  # one 65536 x 65536-bit matrix: 512 MiB
host = np.ones(rows * pitch, dtype=np.uint8)
dev = torch.empty(rows * pitch, dtype=torch.uint8, device="cuda")
H, D = host.ctypes.data, dev.data_ptr()


def best(fn, nbytes):
    fn(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(4):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return nbytes / b / 2 ** 30


print(f"H2D hipMemcpy 512 MiB:                      {best(lambda: hip.hipMemcpy(D, H, rows * pitch, 1), rows * pitch):.1f} GiB/s")
print(f"H2D hipMemcpy2D, rows contiguous (pitch = w): {best(lambda: hip.hipMemcpy2D(D, pitch, H, pitch, pitch, rows, 1), rows * pitch):.1f} GiB/s")
print(f"H2D hipMemcpy2D, half rows (4 KiB of 8 KiB):  {best(lambda: hip.hipMemcpy2D(D, pitch // 2, H, pitch, pitch // 2, rows, 1), rows * pitch // 2):.1f} GiB/s")
print(f"H2D hipMemcpy2D, half the rows, contiguous:   {best(lambda: hip.hipMemcpy2D(D, pitch, H, pitch, pitch, rows // 2, 1), rows * pitch // 2):.1f} GiB/s")
print(f"D2H hipMemcpy 512 MiB:                      {best(lambda: hip.hipMemcpy(H, D, rows * pitch, 2), rows * pitch):.1f} GiB/s")
print(f"D2H hipMemcpy2D, rows contiguous:             {best(lambda: hip.hipMemcpy2D(H, pitch, D, pitch, pitch, rows, 2), rows * pitch):.1f} GiB/s")
print(f"D2H hipMemcpy2D, half rows (quarter of C):    {best(lambda: hip.hipMemcpy2D(H, pitch, D, pitch // 2, pitch // 2, rows // 2, 2), rows * pitch // 4):.1f} GiB/s")
