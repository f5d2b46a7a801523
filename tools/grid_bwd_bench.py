"""hash-grid backward (fill + accumulate) alone on the samples of one 4096-ray batch (event-timed)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.gridencoder import GridEncoder
from laenerf_amd import raymarching as rm
from laenerf_amd.backend import gridencoder_backend as G
dev = "cuda:0"
o, d = S.lego_like_rays(4096, seed=0, n_views=1)
bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
n, f = rm.near_far_from_aabb(to, td, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
c = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, 1.0, bits, 1, 128, n, f, c, -1, True, 128, False, 0, 1024)
enc = GridEncoder(desired_resolution=2048).to(dev)
M = xyzs.shape[0]
grad = (torch.randn(16, M, 2, device=dev) * 1e-2).half()
ge = torch.zeros(enc.embeddings.shape, dtype=torch.half, device=dev)
S_ = np.log2(enc.per_level_scale)
# LAE_BWD_BENCH_TOUCHED=1: with the optimizer's flag word and "ever touched" bitmap, as the training step calls it
extra = {}
if os.environ.get("LAE_BWD_BENCH_TOUCHED"):
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    from laenerf_amd import _lib
    touched = torch.zeros(int(_lib.load().lae_grid_touched_lines_words(enc.embeddings.shape[0])), dtype=torch.int32, device=dev)
    extra = dict(nonfinite_flag=flag.data_ptr(), touched_lines=touched.data_ptr())
def run():
    G.grid_encode_backward(grad, xyzs, None, enc.offsets, ge, M, 3, 2, 16, S_, 16, None, None, 0, False, 0, blc=False, in_map=(1.0, 0.5), offsets_host=enc.offsets_host, **extra)
for rep in range(2):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) / 30 * 1e3:.1f} us (fill + accumulate) for {M} samples")
