"""Drop-in check: an ordinary M4RI client binary (tests/dropin_driver.c, linked against the reference
build) run with LD_PRELOAD=libm4ri_amd.so -- its mzd_mul / mzd_addmul / *_m4rm calls land on the GPU,
everything else (mzd_init, mzd_randomize, mzd_mul_naive, mzd_equal, mzd_free) stays the reference's.
The binary is built in the build container by oracle/Makefile and travels under oracle/_ref/."""
import os
import subprocess

import pytest

import m4ri_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "dropin_driver")


def test_driver_alone_is_a_valid_reference_self_test():
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/dropin_driver not built (needs /root/reference)")
    r = subprocess.run([DRIVER], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "interposed: no" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_ld_preload_drop_in():
    assert os.path.exists(DRIVER), "oracle/_ref/dropin_driver missing: __graft_entry__.build() makes it where /root/reference exists and it travels to the GPU box"
    env = dict(os.environ, LD_PRELOAD=m4ri_amd.LIB_PATH)
    r = subprocess.run([DRIVER], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "interposed: yes" in r.stdout and "ALL OK" in r.stdout and "FAILED" not in r.stdout
    assert "leaf launch" in r.stdout
    # the reference's own TRSM / solve / PLE ran with their internal addmul calls on the GPU
    assert r.stdout.count("  L4 n=") == 2 and "the L4 cases ended on an interposed product" in r.stdout


L4 = os.path.join(ROOT, "oracle", "_ref", "l4_timing_driver")


@pytest.mark.gpu
def test_l4_routines_give_identical_results_under_the_preload():
    """tests/l4_timing_driver.c (M4RI's own mzd_trsm_upper_left / mzd_ple / mzd_solve_left at n = 6000): the
    fingerprints of the TRSM solution and of the PLE decomposition are the same with the products on the
    GPU (preload) as with the reference alone."""
    assert os.path.exists(L4), "oracle/_ref/l4_timing_driver missing: __graft_entry__.build() makes it where /root/reference exists and it travels to the GPU box"
    plain = subprocess.run([L4, "6000"], capture_output=True, text=True, timeout=600)
    pre = subprocess.run([L4, "6000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, LD_PRELOAD=m4ri_amd.LIB_PATH))
    assert plain.returncode == 0 and pre.returncode == 0, plain.stderr[-2000:] + pre.stderr[-2000:]
    assert "interposed: no" in plain.stdout and "interposed: yes" in pre.stdout
    fp = [ln for ln in plain.stdout.splitlines() if "fingerprints:" in ln]
    assert fp and fp == [ln for ln in pre.stdout.splitlines() if "fingerprints:" in ln]
