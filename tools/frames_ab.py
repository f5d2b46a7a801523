"""eval_frame / frame1080 whole + rank-0 shard, short form for A/B runs:  [ENV=...] python tools/frames_ab.py"""
import json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from laenerf_amd import synthetic as S
from laenerf_amd.dist import frame_plan
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
e = bench.eval_frame(dev)
es = bench.eval_frame(dev, density_scale=30.0)
net, r = bench.eval_model(dev, bound=2, seed=1234)
H, W = 1080, 1920
o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
def timed(fn, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts[2:])[len(ts[2:]) // 2] * 1e3
def render(ro, rd, budget=0):
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        return r.render_eval(ro, rd, bg_color=1, max_steps=1024, row_budget=budget, image_hw=(H, W) if ro.shape[0] == H * W else None)
whole = timed(lambda: render(o, d))
whole2 = timed(lambda: render(o, d, 2 * H * W))
take = frame_plan(H * W, 0, 8, dev, (H, W))["take"]
ot, dt = o[take], d[take]
shard = timed(lambda: render(ot, dt, H * W))
shard_ref = timed(lambda: render(ot, dt, 0))
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("LAE_")}, "eval_frame": e["ms_per_frame"], "eval_surface": es["ms_per_frame"],
                  "op_loop_diff": e["max_abs_image_diff_vs_operator_loop"], "f1080": round(whole, 2), "f1080_2N": round(whole2, 2), "shard": round(shard, 3), "shard_ref": round(shard_ref, 3)}))
