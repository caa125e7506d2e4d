#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace CSV: per steady-state step, how much of the wall time the GPU had
(a) at least one kernel running, (b) nothing running (gaps), and the per-queue busy time -- the numbers that say whether a
step is bound by kernel time, by launch gaps on the critical chain, or by the host.

    python tools/trace_timeline.py <kernel_trace.csv> [--steps N] [--marker seed_step_kernel]
A step starts at each launch of the marker kernel (the engine's first kernel of a forward)."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    marker = "seed_step_kernel"
    if "--marker" in sys.argv:
        marker = sys.argv[sys.argv.index("--marker") + 1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) < 3:
        print("not enough steps")
        return
    # last full steps
    for k in range(len(starts) - 3, len(starts) - 1):
        seg = rows[starts[k]:starts[k + 1]]
        t0, t1 = seg[0][0], rows[starts[k + 1]][0]
        # union of busy intervals
        busy = 0
        cur_s, cur_e = seg[0][0], seg[0][1]
        gaps = []
        for s, e, n, q in seg[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, n))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        perq = defaultdict(int)
        for s, e, n, q in seg:
            perq[q] += e - s
        tot = t1 - t0
        print(f"step {k}: wall {tot/1e6:.3f} ms, kernels {len(seg)}, busy(any) {busy/1e6:.3f} ms, idle {(tot-busy)/1e6:.3f} ms, "
              f"sum kernel {sum(e-s for s,e,_,_ in seg)/1e6:.3f} ms")
        print("   per queue busy ms:", {q: round(v / 1e6, 3) for q, v in sorted(perq.items())})
        gaps.sort(reverse=True)
        print("   gap histogram (us): n=%d total=%.3f ms; >20us: %d, 5-20us: %d, <5us: %d" % (
            len(gaps), sum(g for g, _ in gaps) / 1e6, sum(g > 20000 for g, _ in gaps), sum(5000 < g <= 20000 for g, _ in gaps),
            sum(g <= 5000 for g, _ in gaps)))
        print("   largest gaps before:", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:6]])
        # per-kernel-name time in this step
        agg = defaultdict(lambda: [0, 0])
        for s, e, n, q in seg:
            agg[n][0] += e - s
            agg[n][1] += 1
        if "--names" in sys.argv:
            for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
                print(f"      {t/1e3:9.1f} us  x{c:3d}  {n[:110]}")


if __name__ == "__main__":
    main()
