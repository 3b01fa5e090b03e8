"""Measurements bench.py reports BESIDE its number (never inside a timed region): the HBM traffic of one leaf launch from rocprofv3 PMC
passes run over bench.py itself, and the PCIe-inclusive timing of the host entry point."""
from __future__ import annotations

import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

import m4ri_amd

BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def measure_leaf_traffic(argv_size, cutoff, timeout_s=240):
    """HBM bytes of ONE leaf launch of this workload from rocprofv3 PMC passes run right now, on this box,
    over this script in probe mode (one warm-up + one product): FETCH_SIZE and WRITE_SIZE in separate
    passes (they do not fit one), per-dispatch sums over all instances, bytes = (2*FETCH_SIZE + WRITE_SIZE)
    * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read
    stream), WRITE_SIZE as reported.  None (with the reason) when rocprofv3 is unavailable or fails."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="m4ri_amd_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "GRBM_GUI_ACTIVE", "-d", d, "-o", "p", "--", sys.executable, BENCH, "--probe",
               "--size", str(argv_size), "--cutoff", str(cutoff)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("results.db")]
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-300:]}"
            db = sqlite3.connect(dbs[0])
            names = dict(db.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
            disp = {ev: (names.get(kid, ""), end - start) for kid, ev, start, end in
                    db.execute("select kernel_id, event_id, start, end from rocpd_kernel_dispatch")}
            pmc = {pid: nm for pid, nm in db.execute("select id, name from rocpd_info_pmc")}
            per_dispatch, cycles = {}, {}
            for ev, pid, val in db.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
                if "m4rm" not in disp.get(ev, ("", 0))[0]:
                    continue
                if pmc.get(pid) == counter:
                    per_dispatch[ev] = per_dispatch.get(ev, 0.0) + val
                elif pmc.get(pid) == "GRBM_GUI_ACTIVE":
                    cycles[ev] = max(cycles.get(ev, 0.0), val)
            if not per_dispatch:
                return None, f"no {counter} rows for a leaf kernel"
            ev = max(per_dispatch, key=per_dispatch.get)  # the batched leaf launch (strips, if any, are smaller)
            vals[counter] = per_dispatch[ev]
            if ev in cycles and disp[ev][1] > 0:
                vals["gui_cycles"], vals["profiled_ns"] = cycles[ev], disp[ev][1]
        except Exception as e:  # noqa: BLE001
            return None, f"{counter}: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    detail = {"fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"]}
    if "gui_cycles" in vals:  # the clock the launch really ran at (under the profiler): GRBM_GUI_ACTIVE / duration
        detail.update({"gui_active_cycles": vals["gui_cycles"], "profiled_launch_ms": vals["profiled_ns"] * 1e-6,
                       "effective_clock_hz": vals["gui_cycles"] / (vals["profiled_ns"] * 1e-9)})
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, detail


def host_api_timing(A_dev, B_dev, M, L, N, cutoff, runs=3):
    """The other timed region SURVEY.md 8(d) asks for: entry to return of the drop-in host entry point on ordinary
    (pageable) host mzd_t matrices -- H2D of A and B, the device schedule, D2H of C -- next to the device-resident
    number.  Two forms: C given (allocated and touched beforehand) and C == NULL, which is what the reference's
    bench times (bench/bench_multiplication.c:85-107: mzd_mul(NULL, A, B, cutoff) including the allocation of C; here
    the fresh 64-byte-aligned block's pages are first touched by the download).  min / median of `runs` after one warm-up."""
    from m4ri_amd.mzd import Mzd
    lib = m4ri_amd.lib()
    Ah, Bh, Ch = Mzd(M, L), Mzd(L, N), Mzd(M, N)
    Ah.valid_words()[:, :] = A_dev.cpu().numpy().view(np.uint64)   # the very operands of the timed steps, now in host memory
    Bh.valid_words()[:, :] = B_dev.cpu().numpy().view(np.uint64)
    Ch.buf.fill(0)                                                  # touch C's pages: "C given" means a matrix the caller already uses
    out = {}
    lib.mzd_mul(Ch.ptr, Ah.ptr, Bh.ptr, cutoff)                     # warm-up: staging arena, host pipeline threads
    ts = []
    for _ in range(runs):
        t = time.perf_counter()
        lib.mzd_mul(Ch.ptr, Ah.ptr, Bh.ptr, cutoff)
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    out["c_given_ms_min"], out["c_given_ms_median"] = ts[0], ts[len(ts) // 2]
    ts = []
    for _ in range(runs):
        t = time.perf_counter()
        r = lib.mzd_mul(None, Ah.ptr, Bh.ptr, cutoff)
        ts.append((time.perf_counter() - t) * 1e3)
        lib.m4ri_amd_result_free(r)
    ts.sort()
    out["c_null_ms_min"], out["c_null_ms_median"] = ts[0], ts[len(ts) // 2]
    gib = 8.0 * (M * Ah.width + L * Bh.width + M * Ch.width) / 2 ** 30
    out.update({"runs": runs, "GiB_over_pcie": gib, "bitops_per_sec": float(M) * L * N / (out["c_given_ms_min"] * 1e-3),
                "what": "host mzd_mul(C, A, B, cutoff) on pageable mzd_t matrices, PCIe transfers (and for c_null the allocation of C) "
                        "inside the timed region; never the headline `value`"})
    return out
