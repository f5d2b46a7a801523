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

# ---- critical path of an iteration: does the next emit wait for the main stream's compositing or for the lookahead?
look = [(s, s + d) for (n, s, d, g) in rows if 'k_frame_lookahead' in n]
main = [(k, s, s + d) for (n, s, d, g) in rows for k, v in names.items() if v in n]
emits = [i for i, m in enumerate(main) if m[0] == 'emit']
wait_comp = wait_look = 0.0
n_it = 0
gaps = {'emit->grid': 0.0, 'grid->head': 0.0, 'head->comp': 0.0}
for a, b in zip(emits[:-1], emits[1:]):
    grp = main[a:b]
    if [m[0] for m in grp] != ['emit', 'grid', 'head', 'comp']:
        continue
    nxt_start = main[b][1]
    comp_end = grp[3][2]
    la = [l for l in look if grp[0][2] <= l[0] < nxt_start]          # the lookahead launched after this emit
    la_end = la[0][1] if la else comp_end
    wait_comp += (nxt_start - comp_end) / 1e3
    wait_look += max(0.0, (la_end - comp_end)) / 1e3
    gaps['emit->grid'] += (grp[1][1] - grp[0][2]) / 1e3
    gaps['grid->head'] += (grp[2][1] - grp[1][2]) / 1e3
    gaps['head->comp'] += (grp[3][1] - grp[2][2]) / 1e3
    n_it += 1
if n_it:
    print(f"\n{n_it} iterations: next emit starts {wait_comp / n_it:.1f} us after the compositing kernel ends on average; "
          f"the lookahead of the iteration ends {wait_look / n_it:.1f} us AFTER it (0 = earlier); "
          "gaps inside an iteration: " + ", ".join(f"{k} {v / n_it:.1f} us" for k, v in gaps.items()))

la = [(s, d) for (n, s, d, g) in rows if 'k_frame_lookahead' in n]
per_la = len(la) // frames
last_la = la[-per_la:]
print("lookahead durations of the last frame (us), every 5th launch: " + " ".join(f"{d / 1e3:.0f}" for s, d in last_la[::5]))
