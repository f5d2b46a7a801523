"""timeline of ONE replayed train step from a rocprofv3 --kernel-trace CSV of bench.py: main-stream kernels in order with their
durations and the gaps between them, side-stream kernels with their overlap.  python tools/step_timeline.py <kernel_trace.csv>"""
import csv, sys
from collections import Counter
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])))
rows.sort(key=lambda x: x[1])
def short(n):
    for k in ("k_grid_fwd_lean", "k_nerf_head_fwd5", "k_composite_train_step", "k_mlp_bwd_wave", "k_dw_reduce2", "k_bwd_walk", "k_bwd_acc", "k_begin",
              "k_apply_multi", "k_march_train_wave", "k_scan_counts", "k_bwd_scan_units", "k_bwd_scan_parts", "k_near_far"):
        if k in n:
            if k == "k_bwd_walk":
                return "k_bwd_walk<FILL>" if "Lb1ELb1E" in n else "k_bwd_walk<COUNT>"
            return k
    return n[:40]
# the last complete step: from the last-but-one k_grid_fwd_lean to the last
fw = [i for i, r in enumerate(rows) if "k_grid_fwd_lean" in r[0]]
# a replayed step (not one of the eager diagnostic steps at the end): the median-length interval between two encoder launches
iv = sorted((rows[fw[k + 1]][1] - rows[fw[k]][1], k) for k in range(len(fw) // 4, 3 * len(fw) // 4))
k = iv[len(iv) // 2][1]
a, b = fw[k], fw[k + 1]
qmain = rows[a][3]
step = rows[a:b]
t0 = step[0][1]
print(f"step wall {(rows[b][1] - t0) / 1e3:.1f} us")
prev_end = None
gaps = 0.0; ksum = 0.0
for n, s, e, q in step:
    if q != qmain:
        continue
    g = 0.0 if prev_end is None else (s - prev_end) / 1e3
    gaps += g; ksum += (e - s) / 1e3
    print(f"  main {short(n):26s} start {(s - t0) / 1e3:7.1f}  dur {(e - s) / 1e3:6.1f}  gap before {g:5.1f}")
    prev_end = e
print(f"  main kernels {ksum:.1f} us, gaps {gaps:.1f} us, tail gap to next step {(rows[b][1] - prev_end) / 1e3:.1f}")
for n, s, e, q in step:
    if q != qmain:
        print(f"  side {short(n):26s} start {(s - t0) / 1e3:7.1f}  dur {(e - s) / 1e3:6.1f}")
