"""800x800 inference frame: device-resident loop (lae_render_frame) vs the operator-by-operator loop of run_cuda.
Usage: python tools/frame_bench.py [table_amplitude]   (default 1e-4 = untrained grid.py init: every ray marches through)"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = NeRFNetwork(bound=1).to(dev).eval()
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
net.encoder.embeddings.data.uniform_(-amp, amp)
r = NeRFRenderer(net, bound=1).to(dev).eval()
r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
o, d = S.frame_rays(800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
print("rays", tuple(o.shape), "table amplitude", amp)
from laenerf_amd.backend import raymarching_backend as _rb
for name, kw in (("frame_loop", dict(frame_loop=True, want_stats=True)),
                 ("frame_loop no-overlap", dict(frame_loop=True, want_stats=True, _overlap=False)),
                 ("frame_loop budget 4N", dict(frame_loop=True, want_stats=True, row_budget=4 * 640000)),
                 ("op_loop dev-compaction", dict(frame_loop=False)),
                 ("op_loop host mask", dict(frame_loop=False, device_compaction=False))):
    _rb.render_frame_set_overlap(kw.pop("_overlap", True))
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            res = r.render_eval(o, d, bg_color=1, max_steps=1024, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:24s} ms/frame {dt * 1e3:8.2f}  Mrays/s {o.shape[0] / dt / 1e6:7.2f}  hit frac {float((res['weights_sum'] > 0).float().mean()):.3f}",
          res.get("stats", ""))
