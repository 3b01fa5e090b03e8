#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs into the text tables committed under profiles/.

    python tools/rocpd_summary.py <results.db> [...]      # kernel-trace stats and/or PMC sums

Per kernel: calls, total/avg/min/max duration (what `--stats` prints) and, when the db holds
counters, the per-dispatch average of every counter summed over its instances (XCDs/SEs)."""
import sqlite3
import sys
from collections import defaultdict


def summarise(path):
    db = sqlite3.connect(path)
    names = dict(db.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    disp = list(db.execute("select id, kernel_id, start, end, grid_size_x, workgroup_size_x, event_id from rocpd_kernel_dispatch"))
    per = defaultdict(list)
    ev2k = {}
    for (_id, kid, s, e, gx, wx, ev) in disp:
        k = names.get(kid, str(kid)).replace(".kd", "")
        per[k].append((e - s, gx, wx))
        ev2k[ev] = k
    print(f"== {path}")
    total = sum(d for v in per.values() for (d, _, _) in v) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'%':>6s}")
    for k, v in sorted(per.items(), key=lambda kv: -sum(d for d, _, _ in kv[1])):
        ds = [d for d, _, _ in v]
        print(f"{k[:70]:70s} {len(ds):6d} {sum(ds)/1e6:10.3f} {sum(ds)/len(ds)/1e6:10.3f} {min(ds)/1e6:10.3f} {max(ds)/1e6:10.3f} {100*sum(ds)/total:6.1f}")
    pmc_names = dict(db.execute("select id, name from rocpd_info_pmc"))
    rows = list(db.execute("select event_id, pmc_id, value from rocpd_pmc_event"))
    if rows:
        acc = defaultdict(lambda: defaultdict(float))
        ndisp = defaultdict(set)
        for ev, pid, val in rows:
            k = ev2k.get(ev, "?")
            acc[k][pmc_names.get(pid, str(pid))] += val
            ndisp[k].add(ev)
        print("-- counters: per-dispatch average of the sum over all instances")
        for k in acc:
            n = max(1, len(ndisp[k]))
            print(f"  {k[:90]}  ({n} dispatches)")
            for c, v in sorted(acc[k].items()):
                print(f"      {c:28s} {v / n:18.1f}")
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
