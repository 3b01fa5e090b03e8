"""ctypes bindings of the two CPU checkers used by the tests (never by the product path):

  * oracle/liboracle_gf2.so      our plain-C restatement (oracle/gf2_oracle.c), built on demand;
  * oracle/_ref/libm4ri_ref.so   the real reference compiled from /root/reference by oracle/Makefile
                                 (present in the build container and, as a binary, on the GPU box).

All three libraries (these two and libm4ri_amd.so) share one descriptor layout, m4ri_amd.mzd.MzdStruct.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

from m4ri_amd.mzd import Mzd, MzdPtr, from_struct_ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_I = ctypes.c_int


def _stale(target, deps):
    return (not os.path.exists(target)) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


class Oracle:
    def __init__(self, path):
        L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        self.L = L
        sig4 = (MzdPtr, [MzdPtr, MzdPtr, MzdPtr, _I])
        for name, (res, args) in {
            "gf2o_mul": sig4, "gf2o_addmul": sig4, "gf2o_mul_even": sig4, "gf2o_addmul_even": sig4,
            "gf2o_mul_naive": sig4,
            "gf2o_mul_m4rm": (MzdPtr, [MzdPtr, MzdPtr, MzdPtr, _I, _I]),
            "gf2o_add": (MzdPtr, [MzdPtr, MzdPtr, MzdPtr]),
            "gf2o_copy": (MzdPtr, [MzdPtr, MzdPtr]),
            "gf2o_set_zero": (None, [MzdPtr]),
            "gf2o_equal": (_I, [MzdPtr, MzdPtr]),
            "gf2o_fill_splitmix": (None, [MzdPtr, ctypes.c_uint64]),
            "gf2o_fingerprint": (ctypes.c_uint64, [MzdPtr]),
            "gf2o_free": (None, [MzdPtr]),
            "gf2o_trsm_lower_left": (None, [MzdPtr, MzdPtr]),
            "gf2o_trsm_upper_left": (None, [MzdPtr, MzdPtr]),
            "gf2o_trsm_upper_right": (None, [MzdPtr, MzdPtr]),
            "gf2o_trsm_lower_right": (None, [MzdPtr, MzdPtr]),
            "gf2o_ple": (ctypes.c_int32, [MzdPtr, ctypes.c_void_p, ctypes.c_void_p]),
            "gf2o_pluq": (ctypes.c_int32, [MzdPtr, ctypes.c_void_p, ctypes.c_void_p]),
            "gf2o_echelonize": (ctypes.c_int32, [MzdPtr, ctypes.c_int]),
            "gf2o_apply_p_right": (None, [MzdPtr, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
            "gf2o_apply_p_left": (None, [MzdPtr, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
            "gf2o_pluq_solve_left": (ctypes.c_int, [MzdPtr, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, MzdPtr, ctypes.c_int]),
            "gf2o_solve_left": (ctypes.c_int, [MzdPtr, MzdPtr, ctypes.c_int]),
            "gf2o_kernel_left_pluq": (ctypes.c_int32, [MzdPtr, MzdPtr]),
            "gf2o_inv": (None, [MzdPtr, MzdPtr]),
            "gf2o_transpose": (None, [MzdPtr, MzdPtr]),
            "gf2o_trtri_upper": (None, [MzdPtr]),
            "gf2o_ple_recursive": (ctypes.c_int32, [MzdPtr, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
            "gf2o_pluq_recursive": (ctypes.c_int32, [MzdPtr, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
            "gf2o_make_table": (None, [MzdPtr, _I, _I, _I, MzdPtr, ctypes.c_void_p]),
            "gf2o_process_rows": (None, [MzdPtr, _I, _I, _I, _I, _I, ctypes.c_void_p, ctypes.c_void_p]),
        }.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args

    def _ret(self, C, r):
        return C if C is not None else from_struct_ptr(r, self.L.gf2o_free)

    def mul(self, C, A, B, cutoff=0):
        return self._ret(C, self.L.gf2o_mul(C.ptr if C else None, A.ptr, B.ptr, cutoff))

    def addmul(self, C, A, B, cutoff=0):
        return self._ret(C, self.L.gf2o_addmul(C.ptr if C else None, A.ptr, B.ptr, cutoff))

    def mul_naive(self, C, A, B, clear=1):
        self.L.gf2o_mul_naive(C.ptr, A.ptr, B.ptr, clear)
        return C

    def mul_m4rm(self, C, A, B, k=0, clear=1):
        self.L.gf2o_mul_m4rm(C.ptr, A.ptr, B.ptr, k, clear)
        return C

    def add(self, C, A, B):
        self.L.gf2o_add(C.ptr, A.ptr, B.ptr)
        return C

    def trsm_lower_left(self, L, B):
        self.L.gf2o_trsm_lower_left(L.ptr, B.ptr)
        return B

    def trsm_upper_left(self, U, B):
        self.L.gf2o_trsm_upper_left(U.ptr, B.ptr)
        return B

    def make_table(self, M, r, c, k, T, L):
        self.L.gf2o_make_table(M.ptr, r, c, k, T.ptr, L.ctypes.data)

    def process_rows(self, M, startrow, stoprow, startcol, k, Ts, Ls):
        nt = len(Ts)
        tp = (MzdPtr * nt)(*[ctypes.pointer(t.struct) for t in Ts])
        lp = (ctypes.c_void_p * nt)(*[l.ctypes.data for l in Ls])
        self.L.gf2o_process_rows(M.ptr, startrow, stoprow, startcol, k, nt, tp, lp)

    def echelonize(self, A, full):
        return int(self.L.gf2o_echelonize(A.ptr, int(full)))

    def apply_p_right(self, A, P, trans=False):
        import numpy as np
        p = np.ascontiguousarray(P, dtype=np.int32)
        self.L.gf2o_apply_p_right(A.ptr, p.ctypes.data, len(p), int(trans))

    def apply_p_left(self, A, P, trans=False):
        import numpy as np
        p = np.ascontiguousarray(P, dtype=np.int32)
        self.L.gf2o_apply_p_left(A.ptr, p.ctypes.data, len(p), int(trans))

    def solve_left(self, A, B, check=False):
        """A <- its PLUQ, B <- a solution (undefined rows zero); returns 0 or -1 (inconsistent)."""
        return int(self.L.gf2o_solve_left(A.ptr, B.ptr, int(check)))

    def pluq_solve_left(self, A, rank, P, Q, B, check=False):
        import numpy as np
        p, q = np.ascontiguousarray(P, dtype=np.int32), np.ascontiguousarray(Q, dtype=np.int32)
        return int(self.L.gf2o_pluq_solve_left(A.ptr, rank, p.ctypes.data, q.ctypes.data, B.ptr, int(check)))

    def kernel_left_pluq(self, A):
        """A <- its PLUQ; returns (rank, R) with R the ncols x (ncols - rank) kernel basis, or (rank, None)."""
        from m4ri_amd.mzd import Mzd
        r = self.ple(A.copy(), pluq=True, recursive=True)[0]
        R = Mzd(A.ncols, A.ncols - r) if r < A.ncols else None
        got = int(self.L.gf2o_kernel_left_pluq(A.ptr, R.ptr if R is not None else None))
        assert got == r
        return r, R

    def inv(self, A):
        from m4ri_amd.mzd import Mzd
        B = Mzd(A.nrows, A.ncols)
        self.L.gf2o_inv(B.ptr, A.ptr)
        return B

    def transpose(self, A, DST=None):
        from m4ri_amd.mzd import Mzd
        if DST is None:
            DST = Mzd(A.ncols, A.nrows)
        self.L.gf2o_transpose(DST.ptr, A.ptr)
        return DST

    def trtri_upper(self, A):
        self.L.gf2o_trtri_upper(A.ptr)
        return A

    PLE_CUTOFF = 524288  # __M4RI_PLE_CUTOFF (m4ri/ple.h:40) of any build with an L3 of 4 MiB or more

    def ple(self, A, pluq=False, recursive=False, cutoff=PLE_CUTOFF):
        """In place; returns (rank, P, Q) as numpy int32 arrays.  recursive: _mzd_ple / _mzd_pluq (the column-halving
        recursion, ple.c:62-171) instead of _mzd_ple_russian / _mzd_pluq_russian -- same matrix for the PLE, same P,
        same pivots, different leftovers in Q behind the rank."""
        import numpy as np
        P, Q = np.zeros(max(1, A.nrows), dtype=np.int32), np.zeros(max(1, A.ncols), dtype=np.int32)
        if recursive:
            r = (self.L.gf2o_pluq_recursive if pluq else self.L.gf2o_ple_recursive)(A.ptr, P.ctypes.data, Q.ctypes.data, cutoff)
        else:
            r = (self.L.gf2o_pluq if pluq else self.L.gf2o_ple)(A.ptr, P.ctypes.data, Q.ctypes.data)
        return int(r), P[:A.nrows], Q[:A.ncols]

    def trsm_upper_right(self, U, B):
        self.L.gf2o_trsm_upper_right(U.ptr, B.ptr)
        return B

    def trsm_lower_right(self, L_, B):
        self.L.gf2o_trsm_lower_right(L_.ptr, B.ptr)
        return B

    def fill(self, A, seed):
        self.L.gf2o_fill_splitmix(A.ptr, seed)

    def fingerprint(self, A):
        return int(self.L.gf2o_fingerprint(A.ptr))

    def equal(self, A, B):
        return bool(self.L.gf2o_equal(A.ptr, B.ptr))


class Mzp(ctypes.Structure):
    """The reference's mzp_t (m4ri/mzp.h:37-49): LAPACK-style transpositions."""

    _fields_ = [("values", ctypes.POINTER(ctypes.c_int32)), ("length", ctypes.c_int32)]


def mzp_of(arr):
    """An mzp_t over a numpy int32 array (kept alive by the caller)."""
    p = Mzp()
    p.values = arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    p.length = len(arr)
    return p


class Reference:
    """The real M4RI, for pinning the oracle (and as bench.py's cpu_baseline when present)."""

    def __init__(self, path):
        L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        self.L = L
        sig4 = (MzdPtr, [MzdPtr, MzdPtr, MzdPtr, _I])
        sig3 = (MzdPtr, [MzdPtr, MzdPtr, MzdPtr])
        table = {
            "mzd_mul": sig4, "mzd_addmul": sig4, "_mzd_mul_even": sig4, "_mzd_addmul_even": sig4, "_mzd_addmul": sig4,
            "mzd_mul_m4rm": sig4, "mzd_addmul_m4rm": sig4,
            "_mzd_mul_m4rm": (MzdPtr, [MzdPtr, MzdPtr, MzdPtr, _I, _I]),
            "mzd_mul_naive": sig3, "mzd_addmul_naive": sig3, "_mzd_add": sig3,
            "mzd_copy": (MzdPtr, [MzdPtr, MzdPtr]),
            "mzd_equal": (_I, [MzdPtr, MzdPtr]),
            "mzd_free": (None, [MzdPtr]),
            "mzd_init": (MzdPtr, [_I, _I]),
        }
        for name, (res, args) in table.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        for name in ("mzd_trsm_lower_left", "_mzd_trsm_lower_left", "_mzd_trsm_lower_left_russian",
                     "mzd_trsm_upper_left", "_mzd_trsm_upper_left", "_mzd_trsm_upper_left_russian",
                     "mzd_trsm_upper_right", "_mzd_trsm_upper_right", "mzd_trsm_lower_right", "_mzd_trsm_lower_right"):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = None, [MzdPtr, MzdPtr, _I]
        for name in ("_mzd_ple_russian", "_mzd_pluq_russian", "mzd_ple", "mzd_pluq"):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = _I, [MzdPtr, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), _I]
        L.mzd_echelonize.restype, L.mzd_echelonize.argtypes = _I, [MzdPtr, _I]
        L.mzd_echelonize_pluq.restype, L.mzd_echelonize_pluq.argtypes = _I, [MzdPtr, _I]
        L.mzd_echelonize_naive.restype, L.mzd_echelonize_naive.argtypes = _I, [MzdPtr, _I]
        L.mzd_echelonize_m4ri.restype, L.mzd_echelonize_m4ri.argtypes = _I, [MzdPtr, _I, _I]
        L._mzd_echelonize_m4ri.restype, L._mzd_echelonize_m4ri.argtypes = _I, [MzdPtr, _I, _I, _I, ctypes.c_double]
        for name in ("mzd_apply_p_right", "mzd_apply_p_right_trans", "mzd_apply_p_left", "mzd_apply_p_left_trans"):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = None, [MzdPtr, ctypes.POINTER(Mzp)]
        L.mzd_solve_left.restype, L.mzd_solve_left.argtypes = _I, [MzdPtr, MzdPtr, _I, _I]
        L.mzd_pluq_solve_left.restype, L.mzd_pluq_solve_left.argtypes = _I, [MzdPtr, _I, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), MzdPtr, _I, _I]
        L.mzd_kernel_left_pluq.restype, L.mzd_kernel_left_pluq.argtypes = MzdPtr, [MzdPtr, _I]
        L.mzd_inv_m4ri.restype, L.mzd_inv_m4ri.argtypes = MzdPtr, [MzdPtr, MzdPtr, _I]
        L.mzd_transpose.restype, L.mzd_transpose.argtypes = MzdPtr, [MzdPtr, MzdPtr]
        L.mzd_trtri_upper.restype, L.mzd_trtri_upper.argtypes = MzdPtr, [MzdPtr]
        L.mzd_trtri_upper_russian.restype, L.mzd_trtri_upper_russian.argtypes = MzdPtr, [MzdPtr, _I]
        self.has_mp = hasattr(L, "mzd_mul_mp")
        if self.has_mp:
            L.mzd_mul_mp.restype, L.mzd_mul_mp.argtypes = sig4
            L.mzd_addmul_mp.restype, L.mzd_addmul_mp.argtypes = sig4

    def _ret(self, C, r):
        return C if C is not None else from_struct_ptr(r, self.L.mzd_free)

    def mul(self, C, A, B, cutoff=0):
        return self._ret(C, self.L.mzd_mul(C.ptr if C else None, A.ptr, B.ptr, cutoff))

    def addmul(self, C, A, B, cutoff=0):
        return self._ret(C, self.L.mzd_addmul(C.ptr if C else None, A.ptr, B.ptr, cutoff))

    def mul_m4rm(self, C, A, B, k=0):
        return self._ret(C, self.L.mzd_mul_m4rm(C.ptr if C else None, A.ptr, B.ptr, k))

    def addmul_m4rm(self, C, A, B, k=0):
        self.L.mzd_addmul_m4rm(C.ptr, A.ptr, B.ptr, k)
        return C

    def mul_naive(self, C, A, B):
        return self._ret(C, self.L.mzd_mul_naive(C.ptr if C else None, A.ptr, B.ptr))

    def mul_mp(self, C, A, B, cutoff=0):
        return self._ret(C, self.L.mzd_mul_mp(C.ptr if C else None, A.ptr, B.ptr, cutoff))

    def add(self, C, A, B):
        self.L._mzd_add(C.ptr, A.ptr, B.ptr)
        return C

    def echelonize(self, A, full, which="mzd_echelonize_m4ri", k=0):
        if which == "mzd_echelonize_m4ri":
            return int(self.L.mzd_echelonize_m4ri(A.ptr, int(full), k))
        return int(getattr(self.L, which)(A.ptr, int(full)))

    def apply_p(self, A, P, which="mzd_apply_p_right"):
        import numpy as np
        p = np.ascontiguousarray(P, dtype=np.int32)
        getattr(self.L, which)(A.ptr, ctypes.byref(mzp_of(p)))

    def solve_left(self, A, B, check=False):
        return int(self.L.mzd_solve_left(A.ptr, B.ptr, 0, int(check)))

    def pluq_solve_left(self, A, rank, P, Q, B, check=False):
        import numpy as np
        p, q = np.ascontiguousarray(P, dtype=np.int32), np.ascontiguousarray(Q, dtype=np.int32)
        return int(self.L.mzd_pluq_solve_left(A.ptr, rank, ctypes.byref(mzp_of(p)), ctypes.byref(mzp_of(q)), B.ptr, 0, int(check)))

    def kernel_left_pluq(self, A):
        r = self.L.mzd_kernel_left_pluq(A.ptr, 0)
        return from_struct_ptr(r, self.L.mzd_free) if r else None

    def inv(self, A):
        return from_struct_ptr(self.L.mzd_inv_m4ri(None, A.ptr, 0), self.L.mzd_free)

    def transpose(self, A, DST=None):
        return self._ret(DST, self.L.mzd_transpose(DST.ptr if DST is not None else None, A.ptr))

    def trtri_upper(self, A, which="mzd_trtri_upper", k=0):
        if which == "mzd_trtri_upper":
            self.L.mzd_trtri_upper(A.ptr)
        else:
            self.L.mzd_trtri_upper_russian(A.ptr, k)
        return A

    def ple(self, A, which="_mzd_ple_russian", k=0):
        """In place; returns (rank, P, Q)."""
        import numpy as np
        P, Q = np.zeros(max(1, A.nrows), dtype=np.int32), np.zeros(max(1, A.ncols), dtype=np.int32)
        mp, mq = mzp_of(P[:A.nrows] if A.nrows else P), mzp_of(Q[:A.ncols] if A.ncols else Q)
        mp.length, mq.length = A.nrows, A.ncols
        r = getattr(self.L, which)(A.ptr, ctypes.byref(mp), ctypes.byref(mq), k)
        return int(r), P[:A.nrows], Q[:A.ncols]


_oracle = None
_refs = {}


def oracle() -> Oracle:
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, "liboracle_gf2.so")
        deps = [os.path.join(ORACLE_DIR, f) for f in ("gf2_oracle.c", "gf2_oracle.h")]
        if _stale(so, deps):
            subprocess.run(["make", "-C", ORACLE_DIR, "liboracle_gf2.so"], check=True, capture_output=True)
        _oracle = Oracle(so)
    return _oracle


def reference(openmp: bool = False, tag: str = "") -> Reference | None:
    """tag: a build with other cache macros, e.g. "_c32768_1048576_33554432" (oracle/Makefile `ref-cache`)."""
    name = ("libm4ri_ref_omp" if openmp else "libm4ri_ref") + tag + ".so"
    if name not in _refs:
        so = os.path.join(ORACLE_DIR, "_ref", name)
        if not os.path.exists(so) and not tag and os.path.exists("/root/reference/m4ri/mzd.c"):
            subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)
        _refs[name] = Reference(so) if os.path.exists(so) else None
    return _refs[name]
