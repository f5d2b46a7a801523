"""time NeRFRenderer.update_extra_state (renderer.py:555-649): full sweeps (first 16 calls) and partial sweeps"""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
dev = torch.device("cuda:0")
for bound in (1, 2):
    torch.manual_seed(0)
    net = NeRFNetwork(bound=bound).to(dev)
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    r = NeRFRenderer(net, bound=bound, density_thresh=10).to(dev)
    ts = []
    for it in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.float16):
            r.update_extra_state()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"bound {bound} cascades {r.cascade}: full sweep {sorted(ts[2:16])[7]:.2f} ms, partial sweep {sorted(ts[17:])[3]:.2f} ms, "
          f"occupied {float((r.density_grid > 0).float().mean()):.3f}, mean_density {r.mean_density:.3f}")
