#!/usr/bin/env python3
"""Per-kernel medians of one rocprofv3 --pmc pass of SQ counters (see profiles/*_sq_*.txt).
usage: python tools/pmc_sq_summary.py <dir> [substring filters...]"""
import csv
import glob
import os
import statistics
import sys

d = sys.argv[1]
filt = sys.argv[2:] or ["k_mlp_bwd", "k_bwd_walk", "k_bwd_acc", "k_bwd_scan", "k_march", "k_grid_fwd", "k_nerf_head", "k_composite", "k_apply", "k_dw_reduce"]
per = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in filt):
            per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(per):
    c = {n: statistics.median(v) for n, v in per[k].items()}
    print(k[:100])
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    for n in sorted(c):
        print(f"    {n:28s} {c[n]:14.0f}  {100 * c[n] / wc:6.1f} % of SQ_WAVE_CYCLES")
