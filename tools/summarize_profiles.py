#!/usr/bin/env python3
"""Merge the rocprofv3 passes of tools/profile_bench.sh into the tracked summaries under profiles/:

    python tools/summarize_profiles.py gpurun_out/prof_r02 r02

  profiles/<tag>_bench_kernel_stats.csv     the --kernel-trace --stats table (per kernel: calls, total / average ns, share)
  profiles/<tag>_bench_kernel_summary.csv   per kernel: share of GPU time, average duration, HBM bytes per launch from the
                                            PMC passes (2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies 128-B read requests as
                                            64 B, MI355X_MICROARCH.md "HBM"), MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over
                                            the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs [rocprofv3 sums it over the XCDs:
                                            2.77 M for the 0.19 ms LM-head launch = 8 x 346 k cycles] x 256 CUs x 4 SIMDs)
  profiles/kernel_traffic.json              the bytes-per-launch entries bench.py's `roofline.traffic` reads
Only steady-state launches count: for every kernel the first third of its dispatches (bind, warm-up) is dropped."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pat):
    hits = sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))
    return hits[0] if hits else None


def pmc(path):
    """{kernel: {counter: [values in dispatch order]}}"""
    out = defaultdict(lambda: defaultdict(list))
    if not path:
        return out
    with open(path) as f:
        for r in csv.DictReader(f):
            out[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k in out:
        for c in out[k]:
            out[k][c] = [v for _d, v in sorted(out[k][c])]
    return out


def steady_mean(vals):
    if not vals:
        return None
    v = vals[len(vals) // 3:]
    return sum(v) / len(v)


def short(name):
    n = name.replace("void ", "").replace("klab::", "").replace("(anonymous namespace)::", "")
    return n[:110]


def main():
    root, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(here, "profiles")
    stats = find(os.path.join(root, "trace"), "*kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(prof, f"{tag}_bench_kernel_stats.csv"), "w") as f:
        f.write(open(stats).read())
    fetch = pmc(find(os.path.join(root, "fetch"), "*counter_collection.csv"))
    write = pmc(find(os.path.join(root, "write"), "*counter_collection.csv"))
    mf = pmc(find(os.path.join(root, "mfma"), "*counter_collection.csv"))
    out = []
    for r in rows:
        k = r["Name"]
        fb = steady_mean(fetch.get(k, {}).get("FETCH_SIZE", []))
        wb = steady_mean(write.get(k, {}).get("WRITE_SIZE", []))
        busy = steady_mean(mf.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        act = steady_mean(mf.get(k, {}).get("GRBM_GUI_ACTIVE", []))
        hbm = None if fb is None or wb is None else (2.0 * fb + wb) * 1024.0  # counters are in KB
        out.append({"kernel": short(k), "calls": r["Calls"], "share_pct": r["Percentage"], "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                    "fetch_kb_raw": None if fb is None else round(fb, 1), "write_kb": None if wb is None else round(wb, 1),
                    "hbm_mb_per_launch": None if hbm is None else round(hbm / 1e6, 2),
                    "hbm_tb_s": None if hbm is None else round(hbm / (float(r["AverageNs"]) * 1e-9) / 1e12, 2),
                    "mfma_busy_frac": None if not busy or not act else round(busy / (act / 8.0 * 256 * 4), 4), "_full": k})
    with open(os.path.join(prof, f"{tag}_bench_kernel_summary.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=[c for c in out[0] if c != "_full"])
        w.writeheader()
        for o in out:
            w.writerow({c: v for c, v in o.items() if c != "_full"})
    traffic = {}
    for key, pat in (("lmhead", "klab_lmhead_"), ("grouped_wgrad", "grouped_tn_kernel")):
        for o in out:
            if pat in o["_full"] and o["hbm_mb_per_launch"] is not None:
                traffic[key] = {"kernel": o["kernel"], "hbm_bytes_per_launch": int(o["hbm_mb_per_launch"] * 1e6), "fetch_kb_raw": o["fetch_kb_raw"],
                                "write_kb": o["write_kb"], "mfma_busy_frac": o["mfma_busy_frac"],
                                "correction": "gfx950: FETCH_SIZE tallies 128-B read requests as 64 B -> doubled; WRITE_SIZE exact",
                                "source": f"profiles/{tag}_bench_kernel_summary.csv (tools/profile_bench.sh, separate --pmc passes)"}
                break
    json.dump(traffic, open(os.path.join(prof, "kernel_traffic.json"), "w"), indent=1)
    for o in out[:14]:
        print({c: v for c, v in o.items() if c != "_full"})


if __name__ == "__main__":
    main()
