"""Shared cases for the table primitives (mzd_make_table / mzd_process_rowsN): how the reference's own
elimination loop calls them (_mzd_echelonize_m4ri, brilliantrussian.c:603-844: tables from k pivot rows at
(r, c), then every other row processed from column c) plus edge placements."""
import ctypes

import numpy as np

from m4ri_amd.mzd import Mzd

# (nrows, ncols, r, c, k, ntables)
CASES = [(200, 300, 10, 0, 8, 1), (200, 300, 10, 70, 7, 1), (130, 65, 0, 3, 5, 2), (500, 1000, 100, 130, 16, 2), (400, 777, 50, 60, 24, 3),
         (300, 640, 8, 128, 32, 4), (256, 2000, 0, 1000, 40, 5), (700, 4100, 20, 2000, 48, 6), (100, 130, 90, 64, 12, 2), (64, 64, 0, 60, 4, 1),
         (90, 200, 3, 120, 13, 3), (90, 200, 3, 127, 11, 6), (33, 70, 30, 5, 6, 2)]


def split_k(k, n):
    if n == 1:
        return [k]
    if n == 2:
        return [k // 2, k - k // 2]
    rem = k % n
    return [k // n + (1 if (i < n - 1 and rem >= n - 1 - i) else 0) for i in range(n)]


def call_make_table(lib, M, r, c, k, T, L):
    lib.mzd_make_table(M.ptr, r, c, k, T.ptr, L.ctypes.data_as(ctypes.c_void_p))


def call_process_rows(lib, M, startrow, stoprow, startcol, k, Ts, Ls):
    name = "mzd_process_rows" + ("" if len(Ts) == 1 else str(len(Ts)))
    args = []
    for t, l in zip(Ts, Ls):
        args += [t.ptr, l.ctypes.data_as(ctypes.c_void_p)]
    getattr(lib, name)(M.ptr, startrow, stoprow, startcol, k, *args)


def bind_reference(ref):
    from m4ri_amd.mzd import MzdPtr
    _I, _P = ctypes.c_int, ctypes.c_void_p
    L = ref.L
    L.mzd_make_table.restype, L.mzd_make_table.argtypes = None, [MzdPtr, _I, _I, _I, MzdPtr, _P]
    for n in range(1, 7):
        fn = getattr(L, "mzd_process_rows" + ("" if n == 1 else str(n)))
        fn.restype, fn.argtypes = None, [MzdPtr, _I, _I, _I, _I] + [MzdPtr, _P] * n
    return L


def tables_for(make, M, r, c, k, nt):
    """The nt tables the elimination loop would build: table t from rows r + sum(kb[:t]) .. at column c + sum(kb[:t])."""
    kb = split_k(k, nt)
    Ts, Ls, off = [], [], 0
    for t in range(nt):
        T = Mzd.init(1 << kb[t], M.ncols)
        L = np.zeros(1 << kb[t], dtype=np.int32)
        make(M, r + off, c + off, kb[t], T, L)
        Ts.append(T)
        Ls.append(L)
        off += kb[t]
    return Ts, Ls
