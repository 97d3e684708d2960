#!/usr/bin/env python3
"""Summarises a gpurun_prof.sh output directory: per-kernel time (kernel-trace stats) and PMC
counters averaged per launch.  Written next to the raw csv files; copy into profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    for k in ("k_density", "k_force", "k_radix_hist", "k_radix_scan", "k_radix_scatter",
              "k_map_table", "k_reorder", "k_keys", "k_apply_integrate", "k_bodies", "k_pack",
              "k_unpack", "k_bounds", "k_gather4"):
        if k in name:
            return k
    return name[:40]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel-trace stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':22s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{short(r['Name']):22s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:10.1f} "
              f"{float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")

for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
        print("== PMC per launch (mean):", os.path.relpath(f, out))
        for k in acc:
            s = " ".join(f"{c}={acc[k][c]/cnt[k][c]:.4g}" for c in sorted(acc[k]))
            print(f"{k:22s} n={max(cnt[k].values()):4d} {s}")
