#!/usr/bin/env python3
"""The M4RM leaf's design space in LDS-array clocks per inner bit, from the two measured LDS rates (tools/ubench.hip,
profiles/r01_ubench_lds_valu.log: ds_read_b128 = 4 clk per wave-instruction = 256 B/clk/CU; conflict-free ds_write_b128 =
8 clk = 128 B/clk/CU) and the three hard constraints of the C-stationary design:

  * the C tile lives in VGPRs: 512 threads x 128 accumulator dwords = 2 Mbit = R rows x (8 E) columns, E = bytes per table
    entry, so R = 32768 / E;
  * a ds_read_b128 is served in four groups of 16 lanes; random indices are conflict-free only if the 256 / E row groups a
    service group holds read 256 / E DIFFERENT tables laid side by side in one 256-byte bank row.  A resident table set is
    therefore 256 B x 2^k whatever E is: 64 KiB at k = 8, 128 KiB at k = 9, 256 KiB at k = 10;
  * 160 KiB of LDS: two resident sets (build stage s+1 while gathering stage s) fit for k <= 8 only; k = 9 fits once
    (two-phase: build, barrier, gather, barrier -- generation 1's structure), k >= 10 does not fit.

Per stage (one set = 256/E tables = 256 k / E inner bits): gathers R * (256/E) lookups * E bytes / 256 B/clk, table writes
256 B * 2^k / 128 B/clk.  Printed beside it: A bytes a CU pulls per inner bit (R / 8) and per LDS clock, which is what made
generation 5 (E = 32) slower on the box although its LDS-array time fell as predicted.
"""
READ_BPC, WRITE_BPC, LDS_BYTES, TILE_BITS = 256.0, 128.0, 160 * 1024, 512 * 128 * 32

rows = []
for k in (7, 8, 9, 10):
    for E in (256, 128, 64, 32, 16):
        R = TILE_BITS // (8 * E)
        tables = 256 // E
        set_bytes = 256 * (1 << k)
        bits = tables * k
        gather = R * tables * E / READ_BPC
        build = set_bytes / WRITE_BPC
        if 2 * set_bytes <= LDS_BYTES:
            mode, drains = "double-buffered", 0.0
        elif set_bytes <= LDS_BYTES:
            mode, drains = "two-phase", 2 * 150.0   # two full LDS-pipeline drains per stage (generation 1 measured 2 x ~150 clk)
        else:
            mode, drains = "does not fit", None
        if drains is None:
            rows.append((k, E, R, 8 * E, bits, mode, None, None, None))
            continue
        clk = (gather + build + drains) / bits
        rows.append((k, E, R, 8 * E, bits, mode, clk, R / 8.0, R / 8.0 / clk))
base = [r for r in rows if r[0] == 8 and r[1] == 64][0][6]
print(f"{'k':>2} {'entry B':>7} {'tile':>12} {'bits/stage':>10} {'buffers':>16} {'LDS clk/bit':>11} {'vs gen 4':>8} {'A B/bit/CU':>10} {'A B/clk/CU':>10}")
for k, E, R, cols, bits, mode, clk, a_bit, a_clk in rows:
    tile = f"{R}x{cols}"
    if clk is None:
        print(f"{k:>2} {E:>7} {tile:>12} {bits:>10} {mode:>16} {'-':>11} {'-':>8} {'-':>10} {'-':>10}")
    else:
        print(f"{k:>2} {E:>7} {tile:>12} {bits:>10} {mode:>16} {clk:>11.1f} {100 * (clk / base - 1):>+7.1f}% {a_bit:>10.0f} {a_clk:>10.2f}")
