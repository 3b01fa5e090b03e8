#!/usr/bin/env python3
"""Timings of the solvers above the multiply path (TRSM, PLE) with the matrices resident on the device (pinned)
and from host memory.  usage: l4_device_timing.py [n ...]   (PLE_WHICH=_mzd_ple_russian: the flat flavour, also
mzd_pluq / _mzd_pluq_russian; ONLY=ple: skip the triangular solves)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [16384, 32768, 65536]
    m4ri_amd.init(0)
    for n in sizes:
        A, B, T = Mzd.random(n, n, 1), Mzd.random(n, n, 2), Mzd.random(n, n, 3)
        which = os.environ.get("PLE_WHICH", "mzd_ple")
        for what in (("ple",) if os.environ.get("ONLY") == "ple" else ("ple", "trsm_lower", "trsm_upper")):
            for resident in (False, True):
                X = (A if what == "ple" else B).copy()
                if resident:
                    m4ri_amd.pin(X)
                    if what != "ple":
                        m4ri_amd.pin(T)
                t = time.perf_counter()
                if what == "ple":
                    r, P, Q = m4ri_amd.mzd_ple(X, 0, which)
                elif what == "trsm_lower":
                    m4ri_amd.mzd_trsm_lower_left(T, X)
                else:
                    m4ri_amd.mzd_trsm_upper_left(T, X)
                dt = time.perf_counter() - t
                if resident:
                    m4ri_amd.unpin(X)
                    if what != "ple":
                        m4ri_amd.unpin(T)
                extra = f" rank {r} ({which})" if what == "ple" else ""
                print(f"n={n:6d} {what:11s} {'resident' if resident else 'host    '} {dt * 1e3:9.1f} ms{extra}", flush=True)


if __name__ == "__main__":
    main()
