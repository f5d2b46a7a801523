#!/usr/bin/env python3
"""difference of two rocprofv3 `kernel_stats.csv` tables of tools/dropin_profile.py (--steps A and --steps B): calls and device
time per steady drop-in step, per kernel.  usage: dropin_launches.py statsA.csv A statsB.csv B [--json out.json]"""
import csv
import json
import re
import sys


def load(p):
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(k_[a-z0-9_]+?)(I|E)", n)
    if m:
        return m.group(1)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:110]


def main():
    a, ka, b, kb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    d = kb - ka
    rows = []
    for n in b:
        c = (b[n][0] - a.get(n, (0, 0))[0]) / d
        t = (b[n][1] - a.get(n, (0, 0))[1]) / d / 1e3
        if c > 0:
            rows.append((t, c, n))
    rows.sort(reverse=True)
    tot_t, tot_c = sum(r[0] for r in rows), sum(r[1] for r in rows)
    ours = [r for r in rows if short(r[2]).startswith("k_")]
    print(f"per step: {tot_c:.1f} launches, {tot_t:.1f} us of kernels; library kernels {sum(r[1] for r in ours):.1f} launches {sum(r[0] for r in ours):.1f} us; "
          f"torch / runtime kernels {tot_c - sum(r[1] for r in ours):.1f} launches {tot_t - sum(r[0] for r in ours):.1f} us")
    for t, c, n in rows:
        print(f"{t:9.2f} us {c:6.2f} x  {short(n)}")
    if "--json" in sys.argv:
        json.dump({"launches": round(tot_c, 1), "kernel_us": round(tot_t, 1), "library_launches": round(sum(r[1] for r in ours), 1),
                   "library_kernel_us": round(sum(r[0] for r in ours), 1), "source": "rocprofv3 --kernel-trace --stats, difference of a 30-step and a 10-step run "
                   "of tools/dropin_profile.py"}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
