"""Error behaviour of the drop-in entry points: like the reference's m4ri_die (misc.c:36-42) they print
to stderr and abort() -- no error codes.  The argument checks run before any GPU work, so these run on
CPU (each case in a subprocess, because the process dies)."""
import os
import signal
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # strassen.c:346-347
    "mul_inner_mismatch": ("m4ri_amd.mzd_mul(None, Mzd.init(4, 5), Mzd.init(6, 7), 0)", "mzd_mul: A ncols (5) need to match B nrows (6)"),
    # strassen.c:349
    "mul_negative_cutoff": ("m4ri_amd.mzd_mul(None, Mzd.init(4, 5), Mzd.init(5, 7), -1)", "mzd_mul: cutoff must be >= 0"),
    # strassen.c:358-360
    "mul_wrong_c": ("m4ri_amd.mzd_mul(Mzd.init(4, 8), Mzd.init(4, 5), Mzd.init(5, 7), 0)", "mzd_mul: C (4 x 8) has wrong dimensions, expected (4 x 7)"),
    # strassen.c:676-677, :687-690
    "addmul_inner_mismatch": ("m4ri_amd.mzd_addmul(Mzd.init(4, 7), Mzd.init(4, 5), Mzd.init(6, 7), 0)", "mzd_addmul: A ncols (5) need to match B nrows (6)"),
    "addmul_wrong_c": ("m4ri_amd.mzd_addmul(Mzd.init(3, 7), Mzd.init(4, 5), Mzd.init(5, 7), 0)", "mzd_addmul: C (3 x 7) has wrong dimensions"),
    # brilliantrussian.c:1003-1004, :1008-1009
    "m4rm_inner_mismatch": ("m4ri_amd.mzd_mul_m4rm(None, Mzd.init(4, 5), Mzd.init(6, 7), 0)", "mzd_mul_m4rm: A ncols (5) need to match B nrows (6)"),
    "m4rm_wrong_c": ("m4ri_amd.mzd_mul_m4rm(Mzd.init(5, 7), Mzd.init(4, 5), Mzd.init(5, 7), 0)", "mzd_mul_m4rm: C (5 x 7) has wrong dimensions"),
    # triangular.c:396-404, :41-50
    "trsm_lower_left_mismatch": ("m4ri_amd.mzd_trsm_lower_left(Mzd.init(5, 5), Mzd.init(6, 7))", "mzd_trsm_lower_left: L ncols (5) need to match B nrows (6)"),
    "trsm_upper_right_not_square": ("m4ri_amd.lib().mzd_trsm_upper_right(Mzd.init(7, 6).ptr, Mzd.init(5, 7).ptr, 0)", "mzd_trsm_upper_right: U must be square"),
    # ple.c:33-48
    "ple_p_length": ("import ctypes, numpy as np\nA = Mzd.init(4, 5)\nmp, mq = m4ri_amd.Mzp(), m4ri_amd.Mzp()\n"
                     "p, q = np.zeros(3, dtype=np.int32), np.zeros(5, dtype=np.int32)\n"
                     "mp.values, mp.length = p.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 3\nmq.values, mq.length = q.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 5\n"
                     "m4ri_amd.lib().mzd_pluq(A.ptr, ctypes.byref(mp), ctypes.byref(mq), 0)", "mzd_pluq: Permutation P length (3) must match A nrows (4)"),
    # solve.c:30-38
    "solve_left_b_rows": ("m4ri_amd.mzd_solve_left(Mzd.init(4, 6), Mzd.init(5, 3))", "mzd_solve_left: A ncols (6) must be smaller than B nrows (5)"),
    "solve_left_b_rows_max": ("m4ri_amd.mzd_solve_left(Mzd.init(6, 4), Mzd.init(5, 3))", "mzd_solve_left: B nrows (5) must be equal to max of A nrows (6) and A ncols (4)"),
    "inv_not_square": ("m4ri_amd.mzd_inv_m4ri(Mzd.init(4, 5))", "mzd_inv_m4ri: A must be square"),
    # mzd.c:1121-1123
    "transpose_dst_size": ("m4ri_amd.mzd_transpose(Mzd.init(4, 5), Mzd.init(4, 5))", "mzd_transpose: Wrong size for return matrix."),
    # triangular_russian.c:385 (an assert there)
    "trtri_not_square": ("m4ri_amd.mzd_trtri_upper(Mzd.init(4, 5))", "mzd_trtri_upper: matrix must be square"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_fatal_like_m4ri_die(name):
    expr, message = CASES[name]
    code = f"import m4ri_amd\nfrom m4ri_amd.mzd import Mzd\n{expr}\nprint('SURVIVED')\n"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == -signal.SIGABRT, (r.returncode, r.stdout, r.stderr[-500:])
    assert message in r.stderr and "SURVIVED" not in r.stdout


def test_empty_results_need_no_gpu():
    """C with zero rows or columns returns at once (strassen.c:44), and addmul with an empty inner
    dimension leaves C alone (strassen.c:692-695) -- before any device call."""
    code = ("import m4ri_amd\nfrom m4ri_amd.mzd import Mzd\n"
            "c = m4ri_amd.mzd_mul(None, Mzd.init(0, 5), Mzd.init(5, 7), 0); assert (c.nrows, c.ncols) == (0, 7)\n"
            "c = m4ri_amd.mzd_mul(None, Mzd.init(4, 5), Mzd.init(5, 0), 0); assert (c.nrows, c.ncols) == (4, 0)\n"
            "k = Mzd.random(4, 7, 1); c = k.copy(); m4ri_amd.mzd_addmul(c, Mzd.init(4, 0), Mzd.init(0, 7), 0); assert c.equal(k)\n"
            "print('OK')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr[-800:]
