"""per-iteration table of a frame-loop kernel trace: lookahead / finishing kernel against emit / encoder / head, wall per iteration,
and how long the lookahead chain ended after the head kernel (what the next iteration waits for).
python tools/frame_iter_table.py <kernel_trace.csv> [every]"""
import csv, sys
path = sys.argv[1]; every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "rocclr" in n or "at::native" in n:
        continue
    rows.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda x: x[1])
inits = [i for i, r in enumerate(rows) if "k_frame_init" in r[0]]
fins = [i for i, r in enumerate(rows) if "k_frame_finish" in r[0] and "lookahead" not in r[0]]
last = rows[inits[-1]:fins[-1] + 1]
t0 = last[0][1]
sel = lambda key: [r for r in last if key in r[0]]
look, fin, emit, grid, head = sel("k_frame_lookahead<"), sel("k_frame_lookahead_finish"), sel("k_frame_emit"), sel("k_grid_fwd_lean"), sel("k_frame_head")
print(f"frame {(last[-1][2] - t0) / 1e6:.3f} ms; first emit at {(emit[0][1] - t0) / 1e3:.0f} us")
print(" it   look   fin |  emit   grid   head |   wall | lookahead chain ends after head (us)")
tot_wait = 0.0
for i in range(len(emit) - 1):
    lk, fn = look[i + 1], fin[i + 1]
    late = (fn[2] - head[i][2]) / 1e3
    tot_wait += max(late, 0.0)
    if i % every == 0:
        print(f"{i:3d} {(lk[2] - lk[1]) / 1e3:6.1f} {(fn[2] - fn[1]) / 1e3:5.1f} | {(emit[i][2] - emit[i][1]) / 1e3:5.1f} {(grid[i][2] - grid[i][1]) / 1e3:6.1f} {(head[i][2] - head[i][1]) / 1e3:6.1f} | {(emit[i + 1][1] - emit[i][1]) / 1e3:6.1f} | {late:6.1f}")
print(f"sum of positive 'chain ends after head': {tot_wait / 1e3:.2f} ms")
