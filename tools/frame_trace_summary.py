"""summarise a rocprofv3 --kernel-trace CSV of a frame-loop run (tools/frame_prof.py, tools/frame1080_prof.py):
per-kernel time per frame, and for the LAST frame the main-stream chain per iteration (kernel time, gaps between dependent
launches, wall per iteration).  python tools/frame_trace_summary.py <kernel_trace.csv> [frames]"""
import csv, sys
from collections import defaultdict

path = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "rocclr" in n or "at::native" in n:
        continue
    rows.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), int(r["Queue_Id"])))
rows.sort(key=lambda x: x[1])


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("k_frame_lookahead_finish", "k_frame_lookahead", "k_frame_emit", "k_frame_composite", "k_frame_init", "k_frame_finish",
              "k_frame_grid", "k_frame_head", "k_frame_wait", "k_frame_signal", "k_grid_fwd_lean", "k_grid_fwd", "k_nerf_head_fwd5", "k_nerf_head_fwd", "k_near_far"):
        if k in n:
            return k
    return n[:40]


agg = defaultdict(lambda: [0, 0])
for n, s, e, g, q in rows:
    a = agg[short(n)]; a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':28s} {'calls/frame':>11s} {'ms/frame':>9s} {'avg us':>9s} {'%':>6s}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} {c / frames:11.1f} {t / frames / 1e6:9.3f} {t / c / 1e3:9.2f} {100 * t / tot:6.1f}")

# main-stream chain = everything that is not the lookahead pair; iterations are delimited by the first kernel of the chain
inits = [i for i, r in enumerate(rows) if "k_frame_init" in r[0]]
fins = [i for i, r in enumerate(rows) if "k_frame_finish" in r[0] and "lookahead" not in r[0]]
if not inits or not fins:
    sys.exit(0)
lo, hi = inits[-1], fins[-1]
last = rows[lo:hi + 1]
print(f"\nlast frame: {(last[-1][2] - last[0][1]) / 1e6:.3f} ms from k_frame_init start to k_frame_finish end")
# streams by hardware queue: the caller's stream is the one the emit kernel runs on, the lookahead stream the other
q_main = next((r[4] for r in last if "k_frame_emit" in r[0]), None)
q_side = next((r[4] for r in last if "k_frame_lookahead" in r[0] and r[4] != q_main), None)
main = [r for r in last if r[4] == q_main and "lookahead" not in r[0] and "k_frame_init" not in r[0] and "k_frame_finish" not in r[0] and "k_near_far" not in r[0]]
look = [r for r in last if "lookahead" in r[0] or (q_side is not None and r[4] == q_side)]
if not main:
    sys.exit(0)
first = "k_frame_emit"
while main and short(main[0][0]) != first:      # a wait kernel may precede the first emit
    main.pop(0)
its, cur = [], []
for r in main:
    if short(r[0]) == first and cur:
        its.append(cur); cur = []
    cur.append(r)
its.append(cur)
kt = defaultdict(float); gaps = defaultdict(float); wall = 0.0; between = 0.0
for i, it in enumerate(its):
    for j, r in enumerate(it):
        kt[short(r[0])] += (r[2] - r[1]) / 1e3
        if j:
            gaps[f"{short(it[j - 1][0])}->{short(r[0])}"] += (r[1] - it[j - 1][2]) / 1e3
    if i + 1 < len(its):
        wall += (its[i + 1][0][1] - it[0][1]) / 1e3
        between += (its[i + 1][0][1] - it[-1][2]) / 1e3
n = len(its)
print(f"{n} iterations, {wall / max(n - 1, 1):.1f} us wall per iteration; per iteration on the main stream:")
print("  kernels: " + ", ".join(f"{k} {v / n:.1f}" for k, v in kt.items()) + f"  (sum {sum(kt.values()) / n:.1f} us)")
print("  gaps:    " + ", ".join(f"{k} {v / n:.1f}" for k, v in gaps.items()) + f", last->next iteration {between / max(n - 1, 1):.1f}"
      + f"  (sum {(sum(gaps.values()) + between) / n:.1f} us)")
if look:
    la = defaultdict(lambda: [0, 0.0])
    for r in look:
        a = la[short(r[0])]; a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
    print("  side stream: " + ", ".join(f"{k} {v[1] / v[0]:.1f} us x {v[0]}" for k, v in la.items()))
for i in list(range(0, min(6, n))) + list(range(10, n, 20)):
    it = its[i]
    w = (its[i + 1][0][1] - it[0][1]) / 1e3 if i + 1 < n else 0
    print(f"  it {i:3d}: " + " ".join(f"{short(r[0]).replace('k_frame_', '').replace('k_', '')}:{(r[2] - r[1]) / 1e3:6.1f}(g{r[3]})" for r in it) + f" wall {w:.1f}")
