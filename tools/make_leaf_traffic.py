#!/usr/bin/env python3
"""profiles/leaf_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of tools/prof_bench.sh.

usage: make_leaf_traffic.py <pmc_fetch.summary.txt> <pmc_write.summary.txt> <n> <out.json> [source prefix]
Picks the leaf kernel (the m4rm* kernel with the largest counter), bytes per launch =
(2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of
a wide coalesced read stream), WRITE_SIZE as reported; both are per-dispatch sums over all instances.
"""
import json
import re
import sys


def leaf_counter(path, name):
    best, kern, cur = None, None, None
    for line in open(path):
        m = re.match(r"\s+(_Z\S+)\s+\((\d+) dispatches\)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s+" + name + r"\s+([0-9.]+)", line)
        if m and cur and re.search(r"m4rm\w*_kernel", cur):
            v = float(m.group(1))
            if best is None or v > best:
                best, kern = v, cur
    if best is None:
        raise SystemExit(f"{path}: no {name} for a leaf kernel")
    return best, kern


def main():
    fetch, write, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    prefix = sys.argv[5] if len(sys.argv) > 5 else ""
    f, kern = leaf_counter(fetch, "FETCH_SIZE")
    w, _ = leaf_counter(write, "WRITE_SIZE")
    short = re.search(r"(m4rm\w*_kernel)", kern).group(1)
    json.dump({
        "n": n, "kernel": short, "bytes_per_launch": (2 * f + w) * 1024, "fetch_size_kb": f, "write_size_kb": w,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --steps 2 "
                  "--warmup 1` (tools/prof_bench.sh); per-dispatch average summed over all instances; bytes = "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                  "half of a wide coalesced read stream), WRITE_SIZE as reported",
        "source": f"profiles/{prefix}_bench{n}_pmc_fetch.summary.txt, profiles/{prefix}_bench{n}_pmc_write.summary.txt" if prefix else "pmc_fetch.summary.txt, pmc_write.summary.txt",
    }, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
