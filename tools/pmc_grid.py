"""profile target for `rocprofv3 --pmc ... -- python tools/pmc_grid.py`: hash-grid forward launches on one marched lego batch (round 1 counter passes)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.gridencoder import GridEncoder
from laenerf_amd import raymarching as rm
dev = "cuda:0"
o, d = S.lego_like_rays(4096, seed=0)
bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
n, f = rm.near_far_from_aabb(to, td, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
c = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, 1.0, bits, 1, 128, n, f, c, -1, True, 128, False, 0, 1024)
enc = GridEncoder(desired_resolution=2048).to(dev)
print("samples", xyzs.shape)
with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
    for _ in range(5):
        y = enc(xyzs, bound=1)
torch.cuda.synchronize()
