"""summarise a rocprofv3 --kernel-trace run of tools/frame_prof.py (rocpd .db): per-kernel time per frame and the
per-iteration kernel durations of the last frame.  python tools/frame_prof_summary.py <results.db> [frames]"""
import sqlite3, sys
db = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':64s} {'calls':>6s} {'ms/frame':>9s} {'avg us':>9s} {'%':>6s}")
for r in rows[:8]:
    print(f"{r[0][:64]:64s} {r[1]:6d} {r[2] / frames / 1e6:9.3f} {r[3] / 1e3:9.2f} {100 * r[2] / tot:6.1f}")
rows = list(c.execute("select name, start, end-start, grid_x from kernels order by start"))
names = {'emit': 'k_frame_emit', 'grid': 'k_grid_fwd', 'head': 'k_nerf_head', 'comp': 'k_frame_composite'}   # main-stream kernels
seq = [(k, d, g, s) for (n, s, d, g) in rows for k, v in names.items() if v in n]
per = len(seq) // frames
last = seq[-per:]
for it, i in enumerate(range(0, len(last), 4)):
    grp = last[i:i + 4]
    if it < 8 or it % 10 == 0:
        gap = (last[i + 4][3] - grp[0][3]) / 1e3 if i + 4 < len(last) else 0
        print(it, ' '.join(f"{k}:{d / 1e3:6.1f}(g{g})" for k, d, g, s in grp), f"iter wall {gap:.1f}us")
