"""soak test of the device-resident frame loop: many frames of varying size back to back, results checked against the
first render of each size (the loop must neither hang nor depend on host timing)"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = NeRFNetwork(bound=1).to(dev).eval()
net.encoder.embeddings.data.uniform_(-0.5, 0.5)
r = NeRFRenderer(net, bound=1).to(dev).eval()
r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
sizes = [(800, 800), (64, 64), (333, 257), (1080, 1920), (1, 7), (128, 128)]
rays = {}
for hw in sizes:
    o, d = S.frame_rays(*hw)
    rays[hw] = (torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
ref = {}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
rng = np.random.default_rng(0)
for it in range(n):
    hw = sizes[int(rng.integers(len(sizes)))]
    o, d = rays[hw]
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        res = r.render_eval(o, d, bg_color=1, max_steps=1024, row_budget=int(rng.choice([0, 0, 4 * o.shape[0]])))
    if it % 7 == 0:
        torch.cuda.synchronize()                      # sometimes let the queue drain, sometimes pile frames up
    key = hw
    img = res["image"]
    if key not in ref:
        ref[key] = img.clone()
    elif (ref[key] - img).abs().max().item() > 1e-5:
        print("MISMATCH at", it, hw, (ref[key] - img).abs().max().item()); sys.exit(1)
    if it % 50 == 49:
        torch.cuda.synchronize(); print("frames", it + 1, "elapsed", round(time.perf_counter() - t0, 1), "s", flush=True)
torch.cuda.synchronize()
print("soak ok:", n, "frames")
