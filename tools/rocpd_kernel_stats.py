"""per-kernel statistics of a rocprofv3 --kernel-trace run (rocpd .db): calls, total, average, median, min, max (us).
python tools/rocpd_kernel_stats.py <results.db> [--csv out.csv] [--skip N]   (--skip: drop the first N calls of each kernel)"""
import re
import sqlite3
import statistics
import sys

db = sys.argv[1]
csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
c = sqlite3.connect(db)
by = {}
for name, start, end in c.execute("select name, start, end from kernels order by start"):
    by.setdefault(name, []).append((end - start) / 1e3)


def short(n):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(0) if m else n)[:70]


rows = []
for n, v in by.items():
    v = v[skip:] if len(v) > skip else v
    rows.append((short(n), len(v), sum(v), sum(v) / len(v), statistics.median(v), min(v), max(v)))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
hdr = "kernel,calls,total_us,avg_us,median_us,min_us,max_us,percent"
lines = [hdr] + [f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{r[6]:.2f},{100 * r[2] / tot:.1f}" for r in rows]
if csv:
    open(csv, "w").write("\n".join(lines) + "\n")
print(f"{'kernel':70s} {'calls':>6s} {'avg us':>9s} {'median':>9s} {'min':>8s} {'max':>8s} {'%':>6s}")
for r in rows[:40]:
    print(f"{r[0]:70s} {r[1]:6d} {r[3]:9.2f} {r[4]:9.2f} {r[5]:8.2f} {r[6]:8.2f} {100 * r[2] / tot:6.1f}")
