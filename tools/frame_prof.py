"""profile target: N frames with the device-resident loop only (rocprofv3 --kernel-trace --stats -- python tools/frame_prof.py)"""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = NeRFNetwork(bound=1).to(dev).eval()
r = NeRFRenderer(net, bound=1).to(dev).eval()
r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
o, d = S.frame_rays(800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
if os.environ.get("LAE_FRAME_OVERLAP") == "0":              # lookahead in line on the caller's stream (A/B)
    from laenerf_amd.backend import raymarching_backend as _rb
    _rb.render_frame_set_overlap(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for it in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        res = r.render_eval(o, d, bg_color=1, max_steps=1024, want_stats=True, row_budget=budget)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"frame {it}: {dt * 1e3:.2f} ms", res["stats"])
