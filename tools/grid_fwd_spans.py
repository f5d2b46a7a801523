"""per-XCD and per-level busy spans of ONE k_grid_fwd_lean launch (library built with -DLAE_GRID_STAMPS): is the host-built
level -> XCD schedule balanced?  gpurun -- python tools/grid_fwd_spans.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import _lib, synthetic as S, raymarching as rm
from laenerf_amd.gridencoder import GridEncoder
dev = "cuda:0"
o, d = S.lego_like_rays(4096, seed=0, n_views=1)
bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
n, f = rm.near_far_from_aabb(to, td, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
c = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, 1.0, bits, 1, 128, n, f, c, -1, True, 128, False, 0, 1024)
enc = GridEncoder(desired_resolution=2048).to(dev)
enc.embeddings.data.uniform_(-1e-1, 1e-1)
lib = _lib.load()
fn = lib.lae_debug_grid_stamps; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]; fn.restype = ctypes.c_int
with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
    for _ in range(5):
        enc(xyzs, bound=1)
    for rep in range(3):
        enc(xyzs, bound=1)
        buf = np.zeros((8192 * 8, 4), dtype=np.uint64)
        assert fn(buf.ctypes.data, buf.nbytes) == 0
        st = buf.astype(np.int64)
        ok = st[:, 1] > 0
        ok &= st[:, 0] >= st[ok, 0].max() - 100000          # this launch only (1 ms window)
        idx = np.nonzero(ok)[0]
        t0 = st[idx, 0].min()
        xcd = idx & 7
        print(f"launch {rep}: {len(idx)} blocks with work; kernel span {(st[idx, 1].max() - t0) * 0.01:.1f} us; XCD ends (us): "
              + " ".join(f"{(st[idx[xcd == x], 1].max() - t0) * 0.01:5.1f}" for x in range(8)))
        lv = st[idx, 2]
        print("   level [first start..last end] us: " + " ".join(f"{l}:[{(st[idx[lv == l], 0].min() - t0) * 0.01:.0f}..{(st[idx[lv == l], 1].max() - t0) * 0.01:.0f}]" for l in range(16)))
        print("   level on XCD: " + " ".join(f"{l}:{sorted(set((idx[lv == l] & 7).tolist()))}" for l in range(16)))
