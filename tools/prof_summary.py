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
              "k_unpack", "k_bounds", "k_gather4", "k_plan", "k_tile_list"):
        if k in name:
            return k
    return name[:40]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel-trace stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':34s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{short(r['Name']):34s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:10.1f} "
              f"{float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")

# the same trace restricted to the timed region of bench.py: the last steps/(steps+warmup) of each
# kernel's launches (the warm-up steps are launched first)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if steps:
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        per = defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        print(f"== timed region only (last {steps} of {steps + warm} steps):", os.path.relpath(f, out))
        print(f"{'kernel':34s} {'calls':>7s} {'avg_us':>10s} {'ms/step':>10s}")
        for k, v in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
            v.sort()
            m = len(v) // (steps + warm)   # launches per step; a few extra launches may follow
            if m == 0 or len(v) - m * (steps + warm) > 8 * m:
                continue  # not a per-step kernel
            keep = v[m * warm:m * (warm + steps)]
            tot = sum(e - s for s, e in keep)
            print(f"{k:34s} {len(keep):7d} {tot / len(keep) / 1e3:10.1f} {tot / steps / 1e6:10.3f}")

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
            print(f"{k:34s} n={max(cnt[k].values()):4d} {s}")
        # VALU issue utilisation, normalised by SQ_BUSY_CYCLES (summed over the chip's 32 SQ
        # instances, so /32 = the launch's cycles): instructions x measured issue cost (2.6 cycles
        # per wave64 instruction per SIMD with >= 2 waves, profiles/r2_microbench_valu.txt) over
        # 1024 SIMDs x cycles
        lines = []
        for k in acc:
            if "SQ_BUSY_CYCLES" in acc[k] and "SQ_INSTS_VALU" in acc[k]:
                busy = acc[k]["SQ_BUSY_CYCLES"] / cnt[k]["SQ_BUSY_CYCLES"] / 32.0
                valu = acc[k]["SQ_INSTS_VALU"] / cnt[k]["SQ_INSTS_VALU"]
                if busy > 1e5:
                    lines.append(f"{k:34s} cycles={busy:.4g} VALU/SIMD={valu / 1024:.4g} "
                                 f"issue utilisation at 2.6 cyc = {valu / 1024 * 2.6 / busy:.2f}")
        if lines:
            print("== VALU issue utilisation (SQ_BUSY_CYCLES-normalised)")
            print("\n".join(lines))
