"""profile target: configs[3]-shaped 1080p frames with the device-resident loop
(rocprofv3 --kernel-trace -- python3 tools/frame1080_prof.py whole|shard [frames]).
whole: the 1920x1080 frame on one GPU; shard: rank 0's tiles of an 8-way split with the whole frame's row budget
(what each rank of `bench.py --gpus 8` renders before the all-gather)."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from laenerf_amd import synthetic as S
from laenerf_amd.dist import render_shard, render_frame_sharded
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "whole"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
net, r = bench.eval_model(dev, bound=2, seed=1234)
H, W = 1080, 1920
o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
if os.environ.get("LAE_TILE", "8,8") != "0":                 # rays in th x tw pixel tiles (laenerf_amd.dist.PIXEL_TILE, what bench.py renders); "0": scanline order (A/B)
    th, tw = (int(v) for v in os.environ.get("LAE_TILE", "8,8").split(","))
    idx, inv = r._tile_perm(H, W, th, tw, dev)
    o, d = o[idx].contiguous(), d[idx].contiguous()
stats = {}
if os.environ.get("LAE_FRAME_OVERLAP") == "0":              # lookahead in line on the caller's stream (A/B)
    from laenerf_amd.backend import raymarching_backend as _rb
    _rb.render_frame_set_overlap(False)


def render(ro, rd):
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        res = r.render_eval(ro, rd, bg_color=1, max_steps=1024, want_stats=True, row_budget=(H * W if mode == "shard" else 0))
    stats.update(res["stats"])
    return res


for it in range(frames):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "shard":
        render_shard(render, o, d, 0, 8)
    else:
        render_frame_sharded(render, o, d, 0, 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode} frame {it}: {dt * 1e3:.2f} ms", stats, flush=True)
