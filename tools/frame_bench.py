import sys, os, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
dev = torch.device("cuda:0")
net = NeRFNetwork(bound=1).to(dev).eval()
r = NeRFRenderer(net, bound=1).to(dev).eval()
r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
o, d = S.frame_rays(800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
print("rays", o.shape)
for dc in (True, False):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            res = r.render_eval(o, d, bg_color=1, max_steps=1024, device_compaction=dc)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("device_compaction", dc, "ms/frame", round(dt * 1e3, 2), "Mrays/s", round(o.shape[0] / dt / 1e6, 2), "hit frac", float((res["weights_sum"] > 0).float().mean()))
