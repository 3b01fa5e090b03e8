#!/usr/bin/env python3
"""tools/power_trace.py -- socket power, clocks and temperatures of the GPU while a command runs.

    python tools/power_trace.py [--hz 100] [--out DIR] [--tag NAME] -- <command ...>

Samples the amdgpu hwmon files (power*_input / power*_average, freq*_input, temp*_input) of the first GPU at --hz in a thread
while <command> runs as a child, writes DIR/NAME.csv (one row per sample) and DIR/NAME.summary.json: mean / max power, mean and
min shader clock, max temperature over the BUSY samples (power above idle + 25 % of the swing), the power cap, and -- where
`amd-smi` is on the box -- the throttle-residency counters (PPT = package power tracking, thermal, ...) read before and after
the run: their difference is how long the firmware held the clocks down for each reason while the command ran.

Evidence for DESIGN.md 3.1's "the leaf runs on the power limit": not part of the product path.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time


def hip_pci_bus_id(device=0):
    """PCI address of HIP device `device` ("0000:05:00.0"): the box's sysfs shows every GPU of the node, HIP only ours."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            hip = None
    if hip is None:
        return None
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, device) != 0:
        return None
    return buf.value.decode().lower()


def find_hwmon():
    bdf = hip_pci_bus_id(0)
    if bdf:
        hw = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
        cards = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/drm/card*"))
        if hw:
            return (cards[0] if cards else f"/sys/bus/pci/devices/{bdf}"), hw[0], bdf
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        hw = sorted(glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")))
        vendor = os.path.join(card, "device", "vendor")
        try:
            if hw and open(vendor).read().strip() == "0x1002":
                return card, hw[0], None
        except OSError:
            continue
    return None, None, None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def read_text(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def amd_smi(*args, timeout=30):
    exe = "/opt/rocm/bin/amd-smi"
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout)
        return r.stdout if r.returncode == 0 else f"rc {r.returncode}: {r.stderr[-400:]}"
    except Exception as e:  # noqa: BLE001
        return repr(e)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hz", type=float, default=100.0)
    ap.add_argument("--out", default="gpurun_out/power")
    ap.add_argument("--tag", default="trace")
    ap.add_argument("--smi", action="store_true", help="also read amd-smi's throttle / power / clock view before and after")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = args.cmd[1:] if args.cmd and args.cmd[0] == "--" else args.cmd
    if not cmd:
        raise SystemExit("no command")
    os.makedirs(args.out, exist_ok=True)
    card, hw, bdf = find_hwmon()
    if hw is None:
        raise SystemExit("no amdgpu hwmon directory under /sys/class/drm")
    files = {}
    for pat in ("power*_input", "power*_average", "freq*_input", "temp*_input"):
        for p in sorted(glob.glob(os.path.join(hw, pat))):
            name = os.path.basename(p)
            label = read_text(p.rsplit("_", 1)[0] + "_label")
            files[name + (f"[{label}]" if label else "")] = p
    static = {"card": card, "hwmon": hw, "pci_bus_id_of_hip_device_0": bdf, "files": sorted(files),
              "power1_cap_uW": read_int(os.path.join(hw, "power1_cap")), "power1_cap_max_uW": read_int(os.path.join(hw, "power1_cap_max")),
              "power1_cap_default_uW": read_int(os.path.join(hw, "power1_cap_default")),
              "pp_dpm_sclk": read_text(os.path.join(card, "device", "pp_dpm_sclk"))}
    busy_path = os.path.join(card, "device", "gpu_busy_percent")
    smi_before = {"throttle": amd_smi("metric", "--throttle", "--json"), "power": amd_smi("metric", "--power", "--clock", "--temperature", "--json"),
                  "limit": amd_smi("static", "--limit", "--json")} if args.smi else None

    rows, stop = [], threading.Event()
    names = list(files)

    def sampler():
        period = 1.0 / args.hz
        nxt = time.perf_counter()
        while not stop.is_set():
            t = time.perf_counter()
            rows.append((t, *[read_int(files[n]) for n in names], read_int(busy_path)))
            nxt += period
            d = nxt - time.perf_counter()
            if d > 0:
                time.sleep(d)
            else:
                nxt = time.perf_counter()

    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    time.sleep(0.5)  # idle baseline
    t_cmd0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    t_cmd1 = time.perf_counter()
    time.sleep(0.3)
    stop.set()
    th.join()
    smi_after = {"throttle": amd_smi("metric", "--throttle", "--json"), "power": amd_smi("metric", "--power", "--clock", "--temperature", "--json")} if args.smi else None

    csv = os.path.join(args.out, args.tag + ".csv")
    with open(csv, "w") as f:
        f.write("t_s," + ",".join(names) + ",gpu_busy_percent\n")
        for row in rows:
            f.write(f"{row[0] - t0:.4f}," + ",".join("" if v is None else str(v) for v in row[1:]) + "\n")

    def col(name_prefix):
        for i, n in enumerate(names):
            if n.startswith(name_prefix):
                return [row[1 + i] for row in rows if row[1 + i] is not None]
        return []

    power = col("power1_input") or col("power1_average")
    summary = {"cmd": cmd, "rc": r.returncode, "cmd_seconds": t_cmd1 - t_cmd0, "samples": len(rows), "hz_achieved": len(rows) / max(1e-9, rows[-1][0] - rows[0][0]) if len(rows) > 1 else 0,
               "static": static, "stdout_tail": r.stdout[-1500:], "stderr_tail": r.stderr[-500:]}
    if power:
        idle = min(power)
        thr = idle + 0.25 * (max(power) - idle)
        pi = [i for i, n in enumerate(names) if n.startswith("power1_input") or n.startswith("power1_average")][0]
        busy_rows = [row for row in rows if row[1 + pi] is not None and row[1 + pi] >= thr]
        summary["idle_W"] = idle / 1e6
        summary["busy_samples"] = len(busy_rows)
        for i, n in enumerate(names):
            vals = [row[1 + i] for row in busy_rows if row[1 + i] is not None]
            if not vals:
                continue
            scale, unit = (1e-6, "W") if n.startswith("power") else (1e-6, "MHz") if n.startswith("freq") else (1e-3, "C")
            summary[n] = {"unit": unit, "mean": sum(vals) / len(vals) * scale, "min": min(vals) * scale, "max": max(vals) * scale}
    if args.smi:
        summary["amd_smi_before"], summary["amd_smi_after"] = smi_before, smi_after
    with open(os.path.join(args.out, args.tag + ".summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    brief = {k: v for k, v in summary.items() if isinstance(v, dict) and "mean" in v}
    print(json.dumps({"tag": args.tag, "rc": r.returncode, "seconds": round(t_cmd1 - t_cmd0, 2), "cap_W": (static["power1_cap_uW"] or 0) / 1e6,
                      "busy": {k: {kk: round(vv, 1) if isinstance(vv, float) else vv for kk, vv in v.items()} for k, v in brief.items()}}))
    print(r.stdout[-600:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
