#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into
profiles/<tag>_pmc_fetch_write_per_kernel.csv and profiles/grid_fwd_traffic.json.

usage: python tools/pmc_summarize.py <fetch_dir> <write_dir> <tag>
Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM): both counters are KiB; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B, so the read side is doubled (exact for wide streaming reads, an upper estimate for the
4-8 B gathers of the grid encoder, which the guide calls uncalibrated).
"""
import csv
import glob
import json
import os
import statistics
import sys


def medians(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: (len(v), statistics.median(v)) for k, v in per.items()}


def main():
    fetch_dir, write_dir, tag = sys.argv[1:4]
    fe, wr = medians(fetch_dir, "FETCH_SIZE"), medians(write_dir, "WRITE_SIZE")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles", f"{tag}_pmc_fetch_write_per_kernel.csv")
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) of bench.py; medians per dispatch, KiB as reported\n")
        f.write("# gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM)\n")
        f.write("kernel,dispatches,FETCH_SIZE_KiB_median,WRITE_SIZE_KiB_median\n")
        for k in sorted(set(fe) | set(wr)):
            if "at::native" in k or "rocclr" in k:
                continue
            f.write('"%s",%d,%.0f,%.0f\n' % (k, fe.get(k, wr.get(k))[0], fe.get(k, (0, 0))[1], wr.get(k, (0, 0))[1]))
    gk = [k for k in fe if "k_grid_fwd" in k]
    tk = [k for k in fe if "k_out_transpose" in k]
    if gk:
        f_kib = fe[gk[0]][1] + (fe[tk[0]][1] if tk else 0)
        w_kib = wr.get(gk[0], (0, 0))[1] + (wr.get(tk[0], (0, 0))[1] if tk else 0)
        js = {"kernel": "k_grid_fwd + k_out_transpose (one grid_encode_forward call)", "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
              "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024),
              "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024; separate --pmc passes; medians per dispatch; source " + os.path.basename(out),
              "source": "profiles/" + os.path.basename(out)}      # bench.py quotes this in the line's `traffic_source`
        json.dump(js, open(os.path.join(root, "profiles", "grid_fwd_traffic.json"), "w"), indent=1)
        print(js)
    print("wrote", out)


if __name__ == "__main__":
    main()
