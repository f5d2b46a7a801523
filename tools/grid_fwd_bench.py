"""hash-grid forward alone on the samples of one 4096-ray batch (event-timed, 50 launches), for the three forward modes
(lae_grid_set_forward_mode): 0 specialised kernel + balanced XCD schedule, 1 specialised kernel + (l, l+8) map, 2 generic"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S, _lib
from laenerf_amd.gridencoder import GridEncoder
from laenerf_amd import raymarching as rm
from laenerf_amd.backend import gridencoder_backend as G
dev = "cuda:0"
o, d = S.lego_like_rays(4096, seed=0, n_views=int(os.environ.get('VIEWS', '16')))
bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
n, f = rm.near_far_from_aabb(to, td, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
c = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, 1.0, bits, 1, 128, n, f, c, -1, True, 128, False, 0, 1024)
enc = GridEncoder(desired_resolution=2048).to(dev)
table = enc.embeddings.detach().half()
M = xyzs.shape[0]
feats = torch.empty(16, M, 2, dtype=torch.half, device=dev)
def run():
    G.grid_encode_forward(xyzs, table, enc.offsets, feats, M, 3, 2, 16, np.log2(enc.per_level_scale), 16, None, 0, False, 0, blc=False, in_map=(1.0, 0.5))
ref = None
for mode in (2, 1, 0, 0):
    _lib.load().lae_grid_set_forward_mode(mode)
    for _ in range(5): run()
    torch.cuda.synchronize()
    cur = feats.clone()
    if ref is None: ref = cur
    assert torch.equal(ref, cur)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"mode {mode}: {us:.1f} us for {M} samples = {M * 588 / us / 1e3:.0f} GB/s algorithmic")
