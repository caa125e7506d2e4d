#!/usr/bin/env python3
"""List the idle gaps of the busiest queue inside one steady-state step of a rocprofv3 --kernel-trace CSV:
    python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us]
For each gap: its length, the kernel that ended before it and the one that started after it (where the chain waited)."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "seed_step_kernel" in r[2]]
    seg = rows[starts[-2]:starts[-1]]
    byq = defaultdict(list)
    for r in seg:
        byq[r[3]].append(r)
    mainq = max(byq, key=lambda q: sum(e - s for s, e, _n, _q in byq[q]))
    ks = byq[mainq]
    t0 = ks[0][0]
    print(f"step {(rows[starts[-1]][0] - t0) / 1e3:.1f} us, main queue {mainq}: {len(ks)} kernels, busy {sum(e - s for s, e, _n, _q in ks) / 1e3:.1f} us")
    tot = 0.0
    for a, b in zip(ks, ks[1:]):
        g = (b[0] - a[1]) / 1e3
        tot += max(g, 0)
        if g >= min_gap:
            print(f"  t={(a[1] - t0) / 1e3:8.1f} us gap {g:7.1f} us   after {a[2][:60]:60s} before {b[2][:60]}")
    print(f"sum of all gaps on the main queue: {tot:.1f} us")
    for q, v in byq.items():
        if q != mainq:
            print(f"queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, _n, _q in v) / 1e3:.1f} us, first at {(v[0][0] - t0) / 1e3:.1f}, last end {(v[-1][1] - t0) / 1e3:.1f}")


if __name__ == "__main__":
    main()
