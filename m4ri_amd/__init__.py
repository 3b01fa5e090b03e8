"""m4ri_amd -- MI355X-native drop-in for M4RI's dense GF(2) multiply path.

The product is `libm4ri_amd.so` (HIP kernels for gfx950 + the host scheduler + a C ABI that exports
M4RI's own entry points, see include/m4ri_amd.h).  This package is the thin Python binding over that
C ABI: the functions below have M4RI's names and argument meaning (reference m4ri/strassen.h:52-126,
m4ri/brilliantrussian.h:274-317) and simply forward `Mzd` host matrices to the library.

There is no CPU implementation here: if the shared library is missing or there is no GPU, calls
fail loudly.
"""
from __future__ import annotations

import ctypes
import os

from .mzd import Mzd, MzdPtr, MzdStruct, from_struct_ptr, splitmix_words  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libm4ri_amd.so")
_lib = None


class Mzp(ctypes.Structure):
    """mzp_t (include/m4ri_amd.h; reference m4ri/mzp.h:37-49)."""

    _fields_ = [("values", ctypes.POINTER(ctypes.c_int32)), ("length", ctypes.c_int32)]


class ShardPlan(ctypes.Structure):
    """m4ri_amd_shard_plan (include/m4ri_amd.h part 4)."""

    _fields_ = [("world", ctypes.c_int32), ("levels", ctypes.c_int32), ("nprod", ctypes.c_int32), ("blocks", ctypes.c_int32),
                ("m", ctypes.c_int64), ("l", ctypes.c_int64), ("n", ctypes.c_int64),
                ("M", ctypes.c_int64), ("L", ctypes.c_int64), ("N", ctypes.c_int64),
                ("bm", ctypes.c_int64), ("bl", ctypes.c_int64), ("cwl", ctypes.c_int64), ("cwn", ctypes.c_int64)]


class ShardPiece(ctypes.Structure):
    """m4ri_amd_shard_piece."""

    _fields_ = [("holder", ctypes.c_int32), ("owner", ctypes.c_int32),
                ("holder_off", ctypes.c_int64), ("owner_off", ctypes.c_int64), ("words", ctypes.c_int64)]


(BUF_LOCAL_A, BUF_LOCAL_B, BUF_LOCAL_C, BUF_CHILD_A, BUF_CHILD_B, BUF_SLABS_P, BUF_OPER_A, BUF_OPER_B, BUF_PROD) = range(9)


class Stats(ctypes.Structure):
    """m4ri_amd_stats (include/m4ri_amd.h)."""

    _fields_ = [
        ("levels", ctypes.c_int32),
        ("leaf_launches", ctypes.c_int32),
        ("leaf_products", ctypes.c_int64),
        ("leaf_m", ctypes.c_int32),
        ("leaf_l", ctypes.c_int32),
        ("leaf_n", ctypes.c_int32),
        ("leaf_gen", ctypes.c_int32),
        ("leaf_ms", ctypes.c_double),
        ("leaf_bytes", ctypes.c_double),
        ("aux_bytes", ctypes.c_double),
        ("workspace_bytes", ctypes.c_double),
        ("cum_leaf_ms", ctypes.c_double),
        ("cum_leaf_launches", ctypes.c_int64),
    ]


class DmatInfo(ctypes.Structure):
    """m4ri_amd_dmat_info_t."""

    _fields_ = [("rows", ctypes.c_int64), ("ncols", ctypes.c_int64), ("stride", ctypes.c_int64),
                ("layout", ctypes.c_int32), ("world", ctypes.c_int32), ("alive", ctypes.c_int32)]


class MultiStats(ctypes.Structure):
    """m4ri_amd_multi_stats."""

    _fields_ = [("world", ctypes.c_int32), ("variant", ctypes.c_int32), ("levels", ctypes.c_int32), ("sub_products", ctypes.c_int32),
                ("chunks", ctypes.c_int32), ("overlap", ctypes.c_int32), ("converted", ctypes.c_int32), ("pairs_staged", ctypes.c_int32),
                ("m", ctypes.c_int64), ("l", ctypes.c_int64), ("n", ctypes.c_int64), ("link_bytes", ctypes.c_double),
                ("group", ctypes.c_int32), ("reserved", ctypes.c_int32)]


(LAYOUT_ROWS, LAYOUT_CYCLIC1, LAYOUT_CYCLIC2, LAYOUT_REPLICATED) = range(4)
(VARIANT_AUTO, VARIANT_SLABS, VARIANT_STRASSEN) = range(3)
VARIANT_NAMES = {VARIANT_AUTO: "auto", VARIANT_SLABS: "slabs", VARIANT_STRASSEN: "strassen"}

# every symbol include/m4ri_amd.h declares, with its ctypes signature
_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I = ctypes.c_int
_MULSIG = (MzdPtr, [MzdPtr, MzdPtr, MzdPtr, _I])
_DEVSIG = (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I, _I, _P])
SYMBOLS = {
    "mzd_mul": _MULSIG,
    "mzd_addmul": _MULSIG,
    "_mzd_mul_even": _MULSIG,
    "_mzd_addmul_even": _MULSIG,
    "_mzd_addmul": _MULSIG,
    "_mzd_sqr_even": (MzdPtr, [MzdPtr, MzdPtr, _I]),
    "_mzd_addsqr_even": (MzdPtr, [MzdPtr, MzdPtr, _I]),
    "mzd_mul_m4rm": _MULSIG,
    "mzd_addmul_m4rm": _MULSIG,
    "_mzd_mul_m4rm": (MzdPtr, [MzdPtr, MzdPtr, MzdPtr, _I, _I]),
    "mzd_mul_mp": _MULSIG,
    "mzd_addmul_mp": _MULSIG,
    "mzd_trsm_lower_left": (None, [MzdPtr, MzdPtr, _I]),
    "_mzd_trsm_lower_left": (None, [MzdPtr, MzdPtr, _I]),
    "_mzd_trsm_lower_left_russian": (None, [MzdPtr, MzdPtr, _I]),
    "mzd_trsm_upper_left": (None, [MzdPtr, MzdPtr, _I]),
    "_mzd_trsm_upper_left": (None, [MzdPtr, MzdPtr, _I]),
    "_mzd_trsm_upper_left_russian": (None, [MzdPtr, MzdPtr, _I]),
    "mzd_trsm_upper_right": (None, [MzdPtr, MzdPtr, _I]),
    "_mzd_trsm_upper_right": (None, [MzdPtr, MzdPtr, _I]),
    "mzd_trsm_lower_right": (None, [MzdPtr, MzdPtr, _I]),
    "_mzd_trsm_lower_right": (None, [MzdPtr, MzdPtr, _I]),
    "m4ri_amd_trsm_upper_right_dev": (_I, [_P, _I64, _P, _I64, _I64, _I64, _I, _P]),
    "m4ri_amd_trsm_lower_right_dev": (_I, [_P, _I64, _P, _I64, _I64, _I64, _I, _P]),
    "m4ri_amd_trsm_lower_left_dev": (_I, [_P, _I64, _P, _I64, _I64, _I64, _I, _P]),
    "m4ri_amd_trsm_upper_left_dev": (_I, [_P, _I64, _P, _I64, _I64, _I64, _I, _P]),
    "mzd_fprint_row": (None, [_P, MzdPtr, _I]),
    "mzd_fprint": (None, [_P, MzdPtr]),
    "mzd_print": (None, [MzdPtr]),
    "mzd_from_str": (MzdPtr, [_I, _I, ctypes.c_char_p]),
    "mzd_from_jcf": (MzdPtr, [ctypes.c_char_p, _I]),
    "mzd_from_png": (MzdPtr, [ctypes.c_char_p, _I]),
    "mzd_to_png": (_I, [MzdPtr, ctypes.c_char_p, _I, ctypes.c_char_p, _I]),
    "mzd_make_table": (None, [MzdPtr, _I, _I, _I, MzdPtr, _P]),
    "mzd_process_rows": (None, [MzdPtr, _I, _I, _I, _I] + [MzdPtr, _P] * 1),
    "mzd_process_rows2": (None, [MzdPtr, _I, _I, _I, _I] + [MzdPtr, _P] * 2),
    "mzd_process_rows3": (None, [MzdPtr, _I, _I, _I, _I] + [MzdPtr, _P] * 3),
    "mzd_process_rows4": (None, [MzdPtr, _I, _I, _I, _I] + [MzdPtr, _P] * 4),
    "mzd_process_rows5": (None, [MzdPtr, _I, _I, _I, _I] + [MzdPtr, _P] * 5),
    "mzd_process_rows6": (None, [MzdPtr, _I, _I, _I, _I] + [MzdPtr, _P] * 6),
    "m4ri_amd_process_rows_dev": (_I, [_P, _I64, _I64, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _P]),
    "m4ri_amd_make_table_dev": (_I, [_P, _I64, _I64, _I64, _I64, _I64, _I, _P, _P, _I64, _P, _P]),
    "mzd_ple": (_I, [MzdPtr, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), _I]),
    "_mzd_ple": (_I, [MzdPtr, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), _I]),
    "_mzd_ple_russian": (_I, [MzdPtr, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), _I]),
    "mzd_pluq": (_I, [MzdPtr, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), _I]),
    "_mzd_pluq": (_I, [MzdPtr, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), _I]),
    "_mzd_pluq_russian": (_I, [MzdPtr, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), _I]),
    "mzd_apply_p_right_trans_tri": (None, [MzdPtr, ctypes.POINTER(Mzp)]),
    "mzd_apply_p_right": (None, [MzdPtr, ctypes.POINTER(Mzp)]),
    "mzd_apply_p_right_trans": (None, [MzdPtr, ctypes.POINTER(Mzp)]),
    "mzd_apply_p_left": (None, [MzdPtr, ctypes.POINTER(Mzp)]),
    "mzd_apply_p_left_trans": (None, [MzdPtr, ctypes.POINTER(Mzp)]),
    "mzd_solve_left": (_I, [MzdPtr, MzdPtr, _I, _I]),
    "_mzd_solve_left": (_I, [MzdPtr, MzdPtr, _I, _I]),
    "mzd_pluq_solve_left": (_I, [MzdPtr, _I, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), MzdPtr, _I, _I]),
    "_mzd_pluq_solve_left": (_I, [MzdPtr, _I, ctypes.POINTER(Mzp), ctypes.POINTER(Mzp), MzdPtr, _I, _I]),
    "mzd_kernel_left_pluq": (MzdPtr, [MzdPtr, _I]),
    "mzd_inv_m4ri": (MzdPtr, [MzdPtr, MzdPtr, _I]),
    "mzd_echelonize": (_I, [MzdPtr, _I]),
    "mzd_echelonize_m4ri": (_I, [MzdPtr, _I, _I]),
    "mzd_echelonize_pluq": (_I, [MzdPtr, _I]),
    "mzd_echelonize_naive": (_I, [MzdPtr, _I]),
    "_mzd_echelonize_m4ri": (_I, [MzdPtr, _I, _I, _I, ctypes.c_double]),
    "m4ri_amd_ple_dev": (_I, [_P, _I64, _I64, _I64, _P, _P, _P, _I64, _P]),
    "m4ri_amd_pluq_dev": (_I, [_P, _I64, _I64, _I64, _P, _P, _P, _I64, _P]),
    "m4ri_amd_apply_p_right_trans_tri_dev": (_I, [_P, _I64, _I64, _I64, _P, _P]),
    "m4ri_amd_apply_p_left_dev": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I, _P]),
    "m4ri_amd_pluq_solve_left_dev": (_I, [_P, _I64, _I64, _I64, ctypes.c_int32, _P, _P, _P, _I64, _I64, _I64, _I, _I, _P, _P]),
    "m4ri_amd_solve_left_dev": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I, _I, _P, _P]),
    "m4ri_amd_kernel_left_pluq_dev": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I, _P, _P]),
    "m4ri_amd_inv_dev": (_I, [_P, _I64, _P, _I64, _I64, _P]),
    "mzd_transpose": (MzdPtr, [MzdPtr, MzdPtr]),
    "mzd_trtri_upper": (MzdPtr, [MzdPtr]),
    "mzd_trtri_upper_russian": (MzdPtr, [MzdPtr, _I]),
    "m4ri_amd_transpose_dev": (_I, [_P, _I64, _P, _I64, _I64, _I64, _P]),
    "m4ri_amd_m4rm_batch_dev": (_I, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _P]),
    "m4ri_amd_mul_batch_dev": (_I, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _I, _P]),
    "m4ri_amd_model_seconds_batch": (ctypes.c_double, [_I64, _I64, _I64, _I, _I64]),
    "m4ri_amd_trtri_upper_dev": (_I, [_P, _I64, _I64, _P]),
    "m4ri_amd_echelonize_dev": (_I, [_P, _I64, _I64, _I64, _I, _P, _P]),
    "m4ri_amd_apply_p_right_dev": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I, _P]),
    "m4ri_amd_mzd_init": (MzdPtr, [_I, _I]),
    "m4ri_amd_mzd_free": (None, [MzdPtr]),
    "m4ri_amd_result_free": (None, [MzdPtr]),
    "m4ri_amd_init": (_I, [_I]),
    "m4ri_amd_device_count": (_I, []),
    "m4ri_amd_mul_dev": _DEVSIG,
    "m4ri_amd_m4rm_dev": _DEVSIG,
    "m4ri_amd_xor_dev": (_I, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _P]),
    "m4ri_amd_fill_dev": (_I, [_P, _I64, _I64, _I64, ctypes.c_uint64, _P]),
    "m4ri_amd_mask_tail_dev": (_I, [_P, _I64, _I64, _I64, _P]),
    "m4ri_amd_set_profiling": (None, [_I]),
    "m4ri_amd_set_max_fuse": (_I, [_I]),
    "m4ri_amd_plan_levels": (_I, [_I64, _I64, _I64, _I]),
    "m4ri_amd_plan_small_leaf": (_I, [_I64, _I64, _I64, _I64, _I]),
    "m4ri_amd_plan_row_blocks": (_I, [_I64, _I64, _I64, ctypes.c_void_p, ctypes.c_void_p, _I]),
    "m4ri_amd_model_seconds": (ctypes.c_double, [_I64, _I64, _I64, _I]),
    "m4ri_amd_set_workspace_budget": (_I64, [_I64]),
    "m4ri_amd_set_host_pipeline": (_I64, [_I64]),
    "m4ri_amd_pin": (_I, [MzdPtr]),
    "m4ri_amd_sync": (_I, [MzdPtr]),
    "m4ri_amd_host_modified": (_I, [MzdPtr]),
    "m4ri_amd_unpin": (_I, [MzdPtr]),
    "m4ri_amd_is_pinned": (_I, [MzdPtr]),
    "m4ri_amd_get_stats": (_I, [ctypes.POINTER(Stats)]),
    "m4ri_amd_shard_plan_make": (_I, [ctypes.POINTER(ShardPlan), _I, _I64, _I64, _I64, _I]),
    "m4ri_amd_shard_cut": (_I64, [_I64, _I, _I]),
    "m4ri_amd_shard_owner": (_I, [ctypes.POINTER(ShardPlan), _I]),
    "m4ri_amd_shard_group": (_I, [ctypes.POINTER(ShardPlan), _I]),
    "m4ri_amd_shard_slab_rows": (_I64, [ctypes.POINTER(ShardPlan), _I, _I]),
    "m4ri_amd_shard_buffer_words": (_I64, [ctypes.POINTER(ShardPlan), _I, _I]),
    "m4ri_amd_shard_piece_of": (_I, [ctypes.POINTER(ShardPlan), _I, _I, _I, ctypes.POINTER(ShardPiece)]),
    "m4ri_amd_shard_down_dev": (_I, [ctypes.POINTER(ShardPlan), _I, _P, _I64, _P, _I64, _P, _P, _P]),
    "m4ri_amd_shard_up_dev": (_I, [ctypes.POINTER(ShardPlan), _I, _P, _P, _I64, _I, _P]),
    "m4ri_amd_fill_rows_dev": (_I, [_P, _I64, _I64, _I64, _I64, ctypes.c_uint64, _P]),
    "m4ri_amd_mul_multi": (_I, [MzdPtr, MzdPtr, MzdPtr, _I, _I, _I]),
    "m4ri_amd_set_devices": (_I, [_I, ctypes.POINTER(_I)]),
    "m4ri_amd_get_device_list": (_I, [ctypes.POINTER(_I), _I]),
    "m4ri_amd_set_multi_threshold": (_I64, [_I64]),
    "m4ri_amd_release_workspace": (None, []),
    "m4ri_amd_set_small_product_threshold": (_I64, [_I64]),
    "m4ri_amd_small_product_count": (_I64, []),
    "m4ri_amd_small_product_wanted": (_I, [_I64, _I64, _I64]),
    "m4ri_amd_small_mul_host": (_I, [MzdPtr, MzdPtr, MzdPtr, _I]),
    "m4ri_amd_dmat_create": (_P, [_I64, _I64, _I]),
    "m4ri_amd_dmat_free": (None, [_P]),
    "m4ri_amd_dmat_info": (_I, [_P, ctypes.POINTER(DmatInfo)]),
    "m4ri_amd_dmat_local": (_I, [_P, _I, ctypes.POINTER(_P), ctypes.POINTER(_I64), ctypes.POINTER(_I)]),
    "m4ri_amd_dmat_fill": (_I, [_P, ctypes.c_uint64]),
    "m4ri_amd_dmat_upload": (_I, [_P, MzdPtr]),
    "m4ri_amd_dmat_download": (_I, [_P, MzdPtr]),
    "m4ri_amd_dmat_convert": (_I, [_P, _P]),
    "m4ri_amd_dmat_mul": (_I, [_P, _P, _P, _I, _I, _I]),
    "m4ri_amd_dmat_mul_lane": (_I, [_P, _P, _P, _I, _I, _I, _I]),
    "m4ri_amd_multi_sync": (_I, []),
    "m4ri_amd_multi_pair_table": (None, [_I, ctypes.POINTER(ctypes.c_int), _I, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_char_p]),
    "m4ri_amd_multi_link_probe": (_I, [_I64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "m4ri_amd_multi_get_stats": (_I, [ctypes.POINTER(MultiStats)]),
    "m4ri_amd_multi_timeline": (_I, [_I, ctypes.POINTER(ctypes.c_double), _I]),
    "m4ri_amd_multi_default_variant": (_I, [_I, _I64, _I64, _I64]),
    "m4ri_amd_multi_layout_for": (_I, [_I, _I, _I64, _I64, _I64]),
    "m4ri_amd_layout_local_rows": (_I64, [_I, _I, _I, _I64]),
    "m4ri_amd_layout_runs": (_I, [_I, _I, _I, _I64, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64), _I]),
    "m4ri_amd_set_multi_variant": (_I, [_I]),
}


def lib() -> ctypes.CDLL:
    """Load libm4ri_amd.so (RTLD_LOCAL: its M4RI-named symbols must not interpose a libm4ri that
    happens to be loaded in the same process, e.g. the reference build used as the test oracle)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m m4ri_amd.build` (hipcc, gfx950). "
                "m4ri_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _p(m: Mzd | None):
    return m.ptr if m is not None else None


def _ret(C: Mzd | None, r) -> Mzd:
    if C is not None:
        return C
    return from_struct_ptr(r, lib().m4ri_amd_result_free)


# ---- M4RI-named host entry points ---------------------------------------------------------------
def mzd_mul(C: Mzd | None, A: Mzd, B: Mzd, cutoff: int = 0) -> Mzd:
    """C = A*B (strassen.h:52).  C=None allocates.  Fatal (abort) on dimension mismatch, like M4RI."""
    return _ret(C, lib().mzd_mul(_p(C), A.ptr, B.ptr, cutoff))


def mzd_addmul(C: Mzd | None, A: Mzd, B: Mzd, cutoff: int = 0) -> Mzd:
    """C += A*B (strassen.h:68)."""
    return _ret(C, lib().mzd_addmul(_p(C), A.ptr, B.ptr, cutoff))


def _mzd_mul_even(C: Mzd, A: Mzd, B: Mzd, cutoff: int) -> Mzd:
    lib()._mzd_mul_even(C.ptr, A.ptr, B.ptr, cutoff)
    return C


def _mzd_addmul_even(C: Mzd, A: Mzd, B: Mzd, cutoff: int) -> Mzd:
    lib()._mzd_addmul_even(C.ptr, A.ptr, B.ptr, cutoff)
    return C


def _mzd_addmul(C: Mzd, A: Mzd, B: Mzd, cutoff: int) -> Mzd:
    lib()._mzd_addmul(C.ptr, A.ptr, B.ptr, cutoff)
    return C


def mzd_mul_m4rm(C: Mzd | None, A: Mzd, B: Mzd, k: int = 0) -> Mzd:
    """C = A*B by one M4RM leaf, no Strassen (brilliantrussian.h:274).  k is a hint only."""
    return _ret(C, lib().mzd_mul_m4rm(_p(C), A.ptr, B.ptr, k))


def mzd_addmul_m4rm(C: Mzd, A: Mzd, B: Mzd, k: int = 0) -> Mzd:
    lib().mzd_addmul_m4rm(C.ptr, A.ptr, B.ptr, k)
    return C


def _mzd_mul_m4rm(C: Mzd, A: Mzd, B: Mzd, k: int, clear: int) -> Mzd:
    lib()._mzd_mul_m4rm(C.ptr, A.ptr, B.ptr, k, clear)
    return C


def mzd_mul_mp(C: Mzd | None, A: Mzd, B: Mzd, cutoff: int = 0) -> Mzd:
    return _ret(C, lib().mzd_mul_mp(_p(C), A.ptr, B.ptr, cutoff))


def mzd_addmul_mp(C: Mzd | None, A: Mzd, B: Mzd, cutoff: int = 0) -> Mzd:
    return _ret(C, lib().mzd_addmul_mp(_p(C), A.ptr, B.ptr, cutoff))


# ---- triangular solves (reference m4ri/triangular.h:115-153, m4ri/triangular_russian.h:43, :55) -------
def mzd_trsm_lower_left(L: Mzd, B: Mzd, cutoff: int = 0) -> Mzd:
    """B <- L^-1 B in place, L unit lower triangular (only the bits below its diagonal are read)."""
    lib().mzd_trsm_lower_left(L.ptr, B.ptr, cutoff)
    return B


def mzd_trsm_upper_left(U: Mzd, B: Mzd, cutoff: int = 0) -> Mzd:
    """B <- U^-1 B in place, U unit upper triangular (only the bits above its diagonal are read)."""
    lib().mzd_trsm_upper_left(U.ptr, B.ptr, cutoff)
    return B


def mzd_ple(A: Mzd, cutoff: int = 0, which: str = "mzd_ple"):
    """PLE decomposition of A in place (reference m4ri/ple.h:103); returns (rank, P, Q) with P, Q numpy int32
    arrays of A.nrows / A.ncols transpositions."""
    import numpy as np
    P, Q = np.zeros(max(1, A.nrows), dtype=np.int32), np.zeros(max(1, A.ncols), dtype=np.int32)
    mp, mq = Mzp(), Mzp()
    mp.values, mp.length = P.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), A.nrows
    mq.values, mq.length = Q.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), A.ncols
    r = getattr(lib(), which)(A.ptr, ctypes.byref(mp), ctypes.byref(mq), cutoff)
    return int(r), P[:A.nrows], Q[:A.ncols]


def mzd_echelonize(A: Mzd, full: int, which: str = "mzd_echelonize", k: int = 0) -> int:
    """(Reduced) row echelon form of A in place, returns the rank (reference m4ri/echelonform.h:50, :63, :79)."""
    if which == "mzd_echelonize_m4ri":
        return int(lib().mzd_echelonize_m4ri(A.ptr, int(full), k))
    if which == "_mzd_echelonize_m4ri":
        return int(lib()._mzd_echelonize_m4ri(A.ptr, int(full), k, 0, 1.0))
    return int(getattr(lib(), which)(A.ptr, int(full)))


def mzd_apply_p_right(A: Mzd, P, trans: bool = False) -> None:
    """A <- A * P (or A * P^T): the column transpositions (i, P[i]) on every row (reference m4ri/mzp.h:142, :153)."""
    import numpy as np
    p = np.ascontiguousarray(P, dtype=np.int32)
    mp = Mzp()
    mp.values, mp.length = p.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(p)
    (lib().mzd_apply_p_right_trans if trans else lib().mzd_apply_p_right)(A.ptr, ctypes.byref(mp))


def _mzp(values):
    import numpy as np
    v = np.ascontiguousarray(values, dtype=np.int32)
    mp = Mzp()
    mp.values, mp.length = v.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(v)
    return mp, v


def mzd_apply_p_left(A: Mzd, P, trans: bool = False) -> None:
    """The row transpositions (i, P[i]) on A, ascending (or descending: trans) -- reference m4ri/mzp.h:120, :131."""
    mp, keep = _mzp(P)
    (lib().mzd_apply_p_left_trans if trans else lib().mzd_apply_p_left)(A.ptr, ctypes.byref(mp))


def mzd_solve_left(A: Mzd, B: Mzd, cutoff: int = 0, inconsistency_check: bool = False, which: str = "mzd_solve_left") -> int:
    """A X = B in place (reference m4ri/solve.h:50): A <- its PLUQ, B <- X with the undefined rows zero; 0 or -1."""
    return int(getattr(lib(), which)(A.ptr, B.ptr, cutoff, int(inconsistency_check)))


def mzd_pluq_solve_left(A: Mzd, rank: int, P, Q, B: Mzd, cutoff: int = 0, inconsistency_check: bool = False, which: str = "mzd_pluq_solve_left") -> int:
    mp, kp = _mzp(P)
    mq, kq = _mzp(Q)
    return int(getattr(lib(), which)(A.ptr, rank, ctypes.byref(mp), ctypes.byref(mq), B.ptr, cutoff, int(inconsistency_check)))


def mzd_kernel_left_pluq(A: Mzd, cutoff: int = 0):
    """A <- its PLUQ; returns the ncols x (ncols - rank) kernel basis or None (reference m4ri/solve.h:140)."""
    r = lib().mzd_kernel_left_pluq(A.ptr, cutoff)
    return from_struct_ptr(r, lib().m4ri_amd_result_free) if r else None


def mzd_inv_m4ri(A: Mzd, B: Mzd = None) -> Mzd:
    """B <- A^-1 (reference m4ri/brilliantrussian.h:256)."""
    r = lib().mzd_inv_m4ri(B.ptr if B is not None else None, A.ptr, 0)
    return B if B is not None else from_struct_ptr(r, lib().m4ri_amd_result_free)


def mzd_transpose(A: Mzd, DST: Mzd = None) -> Mzd:
    """DST <- A^T (reference m4ri/mzd.h:611)."""
    r = lib().mzd_transpose(DST.ptr if DST is not None else None, A.ptr)
    return DST if DST is not None else from_struct_ptr(r, lib().m4ri_amd_result_free)


def mzd_trtri_upper(A: Mzd, which: str = "mzd_trtri_upper") -> Mzd:
    """A <- A^-1 in place, A unit upper triangular (reference m4ri/triangular.h:163, triangular_russian.h:66)."""
    if which == "mzd_trtri_upper":
        lib().mzd_trtri_upper(A.ptr)
    else:
        lib().mzd_trtri_upper_russian(A.ptr, 0)
    return A


def mzd_apply_p_right_trans_tri(A: Mzd, Q) -> None:
    """Row r of A <- its columns under the transpositions (i, Q[i]), i > r (reference m4ri/mzp.h:202)."""
    import numpy as np
    q = np.ascontiguousarray(Q, dtype=np.int32)
    mq = Mzp()
    mq.values, mq.length = q.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(q)
    lib().mzd_apply_p_right_trans_tri(A.ptr, ctypes.byref(mq))


# ---- I/O formats (reference m4ri/io.h) ---------------------------------------------------------------
def mzd_from_str(m: int, n: int, s: str) -> Mzd:
    return from_struct_ptr(lib().mzd_from_str(m, n, s.encode()), lib().m4ri_amd_result_free)


def mzd_from_jcf(path: str, verbose: int = 0):
    r = lib().mzd_from_jcf(os.fsencode(path), verbose)
    return from_struct_ptr(r, lib().m4ri_amd_result_free) if r else None


def mzd_from_png(path: str, verbose: int = 0):
    r = lib().mzd_from_png(os.fsencode(path), verbose)
    return from_struct_ptr(r, lib().m4ri_amd_result_free) if r else None


def mzd_to_png(A: Mzd, path: str, compression_level: int = -1, comment: str = "", verbose: int = 0) -> int:
    return int(lib().mzd_to_png(A.ptr, os.fsencode(path), compression_level, comment.encode(), verbose))


# ---- device-resident API -------------------------------------------------------------------------
def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"m4ri_amd: {what} failed with hipError_t {rc}")


def init(device: int = 0) -> None:
    _check(lib().m4ri_amd_init(device), "m4ri_amd_init")


def mul_dev(C: int, c_stride: int, A: int, a_stride: int, B: int, b_stride: int, m: int, l: int, n: int,
            add: bool = False, cutoff: int = 0, stream: int = 0) -> None:
    """C (+)= A*B on device pointers (ints), Strassen-Winograd over batched M4RM leaves."""
    _check(lib().m4ri_amd_mul_dev(C, c_stride, A, a_stride, B, b_stride, m, l, n, int(add), cutoff, stream), "m4ri_amd_mul_dev")


def mul_batch_dev(C: int, c_stride: int, c_bs: int, A: int, a_stride: int, a_bs: int, B: int, b_stride: int, b_bs: int, m: int, l: int, n: int,
                  batch: int, add: bool = False, cutoff: int = 0, stream: int = 0) -> None:
    """`batch` products of one shape, X_b = X + b * x_bs words, with the levels of mul_dev and every launch shared by the batch."""
    _check(lib().m4ri_amd_mul_batch_dev(C, c_stride, c_bs, A, a_stride, a_bs, B, b_stride, b_bs, m, l, n, batch, int(add), cutoff, stream),
           "m4ri_amd_mul_batch_dev")


def model_seconds_batch(m: int, l: int, n: int, levels: int = -1, batch: int = 1) -> float:
    """The engine's time model for `batch` products scheduled as one (levels < 0: at the depth the model picks). Host arithmetic."""
    return float(lib().m4ri_amd_model_seconds_batch(m, l, n, levels, batch))


def m4rm_dev(C: int, c_stride: int, A: int, a_stride: int, B: int, b_stride: int, m: int, l: int, n: int,
             add: bool = False, ksplit: int = 0, stream: int = 0) -> None:
    _check(lib().m4ri_amd_m4rm_dev(C, c_stride, A, a_stride, B, b_stride, m, l, n, int(add), ksplit, stream), "m4ri_amd_m4rm_dev")


def xor_dev(C: int, c_stride: int, A: int, a_stride: int, B: int, b_stride: int, rows: int, ncols: int, stream: int = 0) -> None:
    _check(lib().m4ri_amd_xor_dev(C, c_stride, A, a_stride, B, b_stride, rows, ncols, stream), "m4ri_amd_xor_dev")


def fill_dev(M: int, stride: int, rows: int, ncols: int, seed: int, stream: int = 0) -> None:
    _check(lib().m4ri_amd_fill_dev(M, stride, rows, ncols, seed, stream), "m4ri_amd_fill_dev")


def fill_rows_dev(M: int, stride: int, row0: int, rows: int, ncols: int, seed: int, stream: int = 0) -> None:
    """Rows [row0, row0 + rows) of the matrix fill_dev(seed) would produce, written from M's row 0."""
    _check(lib().m4ri_amd_fill_rows_dev(M, stride, row0, rows, ncols, seed, stream), "m4ri_amd_fill_rows_dev")


# ---- several GPUs (include/m4ri_amd.h part 4) ------------------------------------------------------
def shard_plan(world: int, m: int, l: int, n: int, levels: int = 0) -> ShardPlan:
    """Plan of one product over `world` ranks (pure host arithmetic, no GPU needed)."""
    p = ShardPlan()
    if lib().m4ri_amd_shard_plan_make(ctypes.byref(p), world, m, l, n, levels) != 0:
        raise ValueError(f"m4ri_amd_shard_plan_make({world}, {m}, {l}, {n}, {levels}) rejected its arguments")
    return p


def shard_group(plan: ShardPlan, cutoff: int = 0) -> int:
    """Rounds of a rank's sub-products that go into one batched product (mul_batch_dev); 1 = one at a time.  Host arithmetic."""
    return int(lib().m4ri_amd_shard_group(ctypes.byref(plan), cutoff))


def shard_piece(plan: ShardPlan, side: int, j: int, r: int) -> ShardPiece:
    pc = ShardPiece()
    if lib().m4ri_amd_shard_piece_of(ctypes.byref(plan), side, j, r, ctypes.byref(pc)) != 0:
        raise ValueError("m4ri_amd_shard_piece_of: bad arguments")
    return pc


def shard_buffer_words(plan: ShardPlan, rank: int, which: int) -> int:
    return int(lib().m4ri_amd_shard_buffer_words(ctypes.byref(plan), rank, which))


def shard_slab_rows(plan: ShardPlan, rank: int, which: int) -> int:
    return int(lib().m4ri_amd_shard_slab_rows(ctypes.byref(plan), rank, which))


def shard_down_dev(plan: ShardPlan, rank: int, A_local: int, a_stride: int, B_local: int, b_stride: int,
                   child_a: int, child_b: int, stream: int = 0) -> None:
    _check(lib().m4ri_amd_shard_down_dev(ctypes.byref(plan), rank, A_local, a_stride, B_local, b_stride, child_a, child_b, stream),
           "m4ri_amd_shard_down_dev")


def shard_up_dev(plan: ShardPlan, rank: int, slabs_p: int, C_local: int, c_stride: int, add: bool = False, stream: int = 0) -> None:
    _check(lib().m4ri_amd_shard_up_dev(ctypes.byref(plan), rank, slabs_p, C_local, c_stride, int(add), stream), "m4ri_amd_shard_up_dev")


def mul_multi(C: Mzd, A: Mzd, B: Mzd, add: bool = False, cutoff: int = 0, levels: int = 0) -> Mzd:
    """C (+)= A*B over the configured devices (set_devices), host matrices in and out."""
    _check(lib().m4ri_amd_mul_multi(C.ptr, A.ptr, B.ptr, int(add), cutoff, levels), "m4ri_amd_mul_multi")
    return C


def set_devices(ids) -> None:
    """Devices mzd_mul_mp / mul_multi spread a product over; an id may repeat (ranks sharing a GPU);
    an empty list restores the default (M4RI_AMD_DEVICES or every visible device)."""
    arr = (ctypes.c_int * max(1, len(ids)))(*ids)
    if lib().m4ri_amd_set_devices(len(ids), arr) != 0:
        raise ValueError(f"m4ri_amd_set_devices({list(ids)}): unknown device id")


def get_devices() -> list:
    arr = (ctypes.c_int * 64)()
    n = lib().m4ri_amd_get_device_list(arr, 64)
    return [int(arr[i]) for i in range(min(n, 64))]


def set_multi_threshold(min_dim: int) -> int:
    return int(lib().m4ri_amd_set_multi_threshold(int(min_dim)))


# ---- residency (include/m4ri_amd.h part 3) --------------------------------------------------------
class Dmat:
    """A matrix distributed over the configured devices and resident in HBM (m4ri_amd_dmat, include/m4ri_amd.h part 4)."""

    def __init__(self, rows: int, ncols: int, layout: int = LAYOUT_ROWS):
        self.h = lib().m4ri_amd_dmat_create(rows, ncols, layout)
        if not self.h:
            raise RuntimeError(f"m4ri_amd_dmat_create({rows}, {ncols}, {layout}) failed")
        self.rows, self.ncols, self.layout = rows, ncols, layout

    def free(self):
        if self.h:
            lib().m4ri_amd_dmat_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def info(self) -> DmatInfo:
        out = DmatInfo()
        _check(lib().m4ri_amd_dmat_info(self.h, ctypes.byref(out)), "m4ri_amd_dmat_info")
        return out

    def local(self, rank: int):
        """(device pointer, rows incl. padding, HIP device) of rank `rank`'s local buffer."""
        ptr, rows, dev = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int()
        _check(lib().m4ri_amd_dmat_local(self.h, rank, ctypes.byref(ptr), ctypes.byref(rows), ctypes.byref(dev)), "m4ri_amd_dmat_local")
        return ptr.value, rows.value, dev.value

    def fill(self, seed: int) -> "Dmat":
        _check(lib().m4ri_amd_dmat_fill(self.h, seed), "m4ri_amd_dmat_fill")
        return self

    def upload(self, M: Mzd) -> "Dmat":
        _check(lib().m4ri_amd_dmat_upload(self.h, M.ptr), "m4ri_amd_dmat_upload")
        return self

    def download(self, M: Mzd = None) -> Mzd:
        M = M if M is not None else Mzd(self.rows, self.ncols)
        _check(lib().m4ri_amd_dmat_download(self.h, M.ptr), "m4ri_amd_dmat_download")
        return M

    def convert_from(self, src: "Dmat") -> "Dmat":
        _check(lib().m4ri_amd_dmat_convert(self.h, src.h), "m4ri_amd_dmat_convert")
        return self


def dmat_mul(C: Dmat, A: Dmat, B: Dmat, add: bool = False, cutoff: int = 0, variant: int = VARIANT_AUTO, lane: int = 0) -> Dmat:
    """C (+)= A*B on distributed operands; asynchronous (multi_sync).  lane 0 / 1: independent products issued alternately on the two
    lanes keep two in flight (m4ri_amd_dmat_mul_lane)."""
    if lane:
        _check(lib().m4ri_amd_dmat_mul_lane(C.h, A.h, B.h, int(add), cutoff, variant, lane), "m4ri_amd_dmat_mul_lane")
    else:
        _check(lib().m4ri_amd_dmat_mul(C.h, A.h, B.h, int(add), cutoff, variant), "m4ri_amd_dmat_mul")
    return C


def multi_pair_table(devices, can_access, no_peer: str = "") -> list:
    """The pure rule behind the multi-device path's pair table (no GPU): staged[dst][src] = 1 where copies dst <- src go through the
    host.  `can_access`: ndev x ndev nested lists (hipDeviceCanAccessPeer)."""
    w, ndev = len(devices), len(can_access)
    dev = (ctypes.c_int * w)(*devices)
    can = (ctypes.c_int * (ndev * ndev))(*[int(x) for row in can_access for x in row])
    out = ctypes.create_string_buffer(w * w)
    lib().m4ri_amd_multi_pair_table(w, dev, ndev, can, no_peer.encode() if no_peer else None, out)
    return [[out.raw[i * w + j] for j in range(w)] for i in range(w)]


def multi_link_probe(nbytes: int = 256 << 20) -> dict:
    """Every ordered pair of ranks copies `nbytes` one pair at a time, then all pairs at once (m4ri_amd_multi_link_probe)."""
    w = len(get_devices())
    pair = (ctypes.c_double * (w * w))()
    allg, same = ctypes.c_double(0.0), ctypes.c_int(0)
    staged = ctypes.create_string_buffer(w * w)
    _check(lib().m4ri_amd_multi_link_probe(nbytes, pair, ctypes.byref(allg), staged, ctypes.byref(same)), "m4ri_amd_multi_link_probe")
    rates = sorted(pair[i * w + j] for i in range(w) for j in range(w) if i != j)
    return {"bytes_per_copy": nbytes, "pairs": len(rates), "gbs_per_direction_min": rates[0] if rates else None,
            "gbs_per_direction_median": rates[len(rates) // 2] if rates else None, "gbs_per_direction_max": rates[-1] if rates else None,
            "all_at_once_gbs": allg.value, "all_at_once_gbs_per_direction": allg.value / max(1, len(rates)),
            "pair_gbs": [[round(pair[i * w + j], 2) for j in range(w)] for i in range(w)],
            "peer_access": [[int(i == j or not staged.raw[i * w + j]) for j in range(w)] for i in range(w)],
            "ranks_share_devices": bool(same.value)}


def multi_sync() -> None:
    _check(lib().m4ri_amd_multi_sync(), "m4ri_amd_multi_sync")


def multi_stats() -> MultiStats:
    out = MultiStats()
    _check(lib().m4ri_amd_multi_get_stats(ctypes.byref(out)), "m4ri_amd_multi_get_stats")
    return out


def multi_timeline(rank: int, cap: int = 256) -> list:
    buf = (ctypes.c_double * cap)()
    n = lib().m4ri_amd_multi_timeline(rank, buf, cap)
    if n < 0:
        raise RuntimeError("m4ri_amd_multi_timeline failed")
    return [buf[i] for i in range(n)]


def multi_default_variant(world: int, m: int, l: int, n: int) -> int:
    return lib().m4ri_amd_multi_default_variant(world, m, l, n)


def multi_layout_for(variant: int, world: int, m: int, l: int, n: int) -> int:
    return lib().m4ri_amd_multi_layout_for(variant, world, m, l, n)


def layout_runs(layout: int, world: int, rank: int, rows: int) -> list:
    """[(global first row, rows, local first row)] of the valid rows rank `rank` holds."""
    g0, nr, l0 = (ctypes.c_int64 * 4)(), (ctypes.c_int64 * 4)(), (ctypes.c_int64 * 4)()
    n = lib().m4ri_amd_layout_runs(layout, world, rank, rows, g0, nr, l0, 4)
    if n < 0:
        raise ValueError("m4ri_amd_layout_runs: bad arguments")
    return [(g0[i], nr[i], l0[i]) for i in range(n)]


def set_multi_variant(variant: int) -> int:
    return lib().m4ri_amd_set_multi_variant(variant)


def set_small_product_threshold(ops: int) -> int:
    """m * l * n at or below which a product from host memory is computed on the host (0: never); returns the previous value."""
    return int(lib().m4ri_amd_set_small_product_threshold(int(ops)))


def small_product_count() -> int:
    return int(lib().m4ri_amd_small_product_count())


def small_mul_host(C: Mzd, A: Mzd, B: Mzd, add: bool = False) -> Mzd:
    """The host Four Russians of the small-product path, called directly (no device involved)."""
    if lib().m4ri_amd_small_mul_host(C.ptr, A.ptr, B.ptr, int(add)) != 0:
        raise ValueError("m4ri_amd_small_mul_host: mismatched dimensions")
    return C


def pin(M: Mzd) -> None:
    """Keep a device copy of M (which must own its block); products then read it, and windows into it,
    in place and leave results there.  The host copy is stale after a product wrote into it until sync()."""
    if lib().m4ri_amd_pin(M.ptr) != 0:
        raise ValueError("m4ri_amd_pin: matrix is a window, empty or NULL")


def sync(M: Mzd) -> None:
    if lib().m4ri_amd_sync(M.ptr) != 0:
        raise ValueError("m4ri_amd_sync: matrix is not pinned")


def host_modified(M: Mzd) -> None:
    if lib().m4ri_amd_host_modified(M.ptr) != 0:
        raise ValueError("m4ri_amd_host_modified: matrix is not pinned")


def unpin(M: Mzd) -> None:
    if lib().m4ri_amd_unpin(M.ptr) != 0:
        raise ValueError("m4ri_amd_unpin: matrix is not pinned")


def is_pinned(M: Mzd) -> int:
    return int(lib().m4ri_amd_is_pinned(M.ptr))


def plan_row_blocks(m: int, l: int, n: int):
    """The engine's own plan (cutoff 0): [(rows, levels), ...], largest block of rows first (m4ri_amd_plan_row_blocks)."""
    rows = (ctypes.c_int64 * 16)()
    levels = (ctypes.c_int * 16)()
    k = int(lib().m4ri_amd_plan_row_blocks(m, l, n, ctypes.cast(rows, ctypes.c_void_p), ctypes.cast(levels, ctypes.c_void_p), 16))
    return [(int(rows[i]), int(levels[i])) for i in range(min(k, 16))]


def model_seconds(m: int, l: int, n: int, levels: int) -> float:
    """The engine's time model for one product at a depth (m4ri_amd_model_seconds)."""
    return float(lib().m4ri_amd_model_seconds(m, l, n, levels))


def plan_levels(m: int, l: int, n: int, cutoff: int = 0) -> int:
    """Strassen-Winograd levels the engine would use (host logic only, no GPU needed)."""
    return int(lib().m4ri_amd_plan_levels(m, l, n, cutoff))


def set_workspace_budget(nbytes: int) -> int:
    """Bytes the breadth-first workspace may take (0 = automatic); returns the previous value."""
    return int(lib().m4ri_amd_set_workspace_budget(int(nbytes)))


def set_host_pipeline(min_bytes: int) -> int:
    """A + B + C bytes from which host-memory products are pipelined over row slabs (0 = never); returns the previous value."""
    return int(lib().m4ri_amd_set_host_pipeline(int(min_bytes)))


def set_max_fuse(levels: int) -> int:
    """Deepest Strassen-Winograd levels covered by one fused pass (1..4); returns the previous value."""
    return int(lib().m4ri_amd_set_max_fuse(int(levels)))


def set_profiling(on) -> None:
    """0/False off, 1/True per product, 2 cumulative over products (Stats.cum_leaf_ms / cum_leaf_launches)."""
    lib().m4ri_amd_set_profiling(int(on))


def get_stats() -> Stats:
    s = Stats()
    _check(lib().m4ri_amd_get_stats(ctypes.byref(s)), "m4ri_amd_get_stats")
    return s
