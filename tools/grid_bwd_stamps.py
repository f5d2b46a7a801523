"""where a block of the hash-grid backward spends its life: phase stamps of k_bwd_walk<FILL> and k_bwd_acc (100 MHz wall clock).

Needs the library built with the stamps compiled in (they are not in the shipped build):
    LAE_BUILD_EXTRA_FLAGS=-DLAE_GRID_STAMPS python -m laenerf_amd.build --force
    gpurun -- python tools/grid_bwd_stamps.py
    python -m laenerf_amd.build --force          # back to the shipped library
Same samples as tools/grid_bwd_bench.py (one 4096-ray batch of the bench scene)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from laenerf_amd import _lib                                  # noqa: E402
from laenerf_amd import raymarching as rm                     # noqa: E402
from laenerf_amd import synthetic as S                        # noqa: E402
from laenerf_amd.backend import gridencoder_backend as G      # noqa: E402
from laenerf_amd.gridencoder import GridEncoder               # noqa: E402

dev = "cuda:0"
if os.environ.get("LAE_STAMPS_SCENE") == "style":              # bench.py style_step's points: 100 k random points in a 0.3-radius ball
    torch.manual_seed(7)
    v = torch.randn(100000, 3, device=dev)
    xyzs = (v / v.norm(dim=-1, keepdim=True) * 0.3 * torch.rand(100000, 1, device=dev) ** (1 / 3)).contiguous()
else:
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


def run():
    G.grid_encode_backward(grad, xyzs, None, enc.offsets, ge, M, 3, 2, 16, S_, 16, None, None, 0, False, 0, blc=False,
                           in_map=(1.0, 0.5), offsets_host=enc.offsets_host)


for _ in range(5):
    run()
torch.cuda.synchronize()
lib = _lib.load()
fn = lib.lae_debug_grid_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
fn.restype = ctypes.c_int
buf = np.zeros((8192, 32), dtype=np.uint64)
assert fn(buf.ctypes.data, buf.nbytes) == 0
st = buf.astype(np.int64)
TICK = 0.01                                                   # us per tick

L = 16
U = (M + 1023) // 1024
nblk = U * L
fill = st[:nblk]
t0 = fill[:, 0].min()
print(f"{M} samples, {U} units per level, {nblk} fill blocks")
print(f"FILL kernel span (first block start -> last block end): {(fill[:, 6].max() - t0) * TICK:.1f} us")
life = (fill[:, 6] - fill[:, 0]) * TICK
print(f"block life: mean {life.mean():.2f} us, median {np.median(life):.2f}, p90 {np.percentile(life, 90):.2f}, max {life.max():.2f};"
      f" sum of lives / span = {life.sum() / ((fill[:, 6].max() - t0) * TICK):.0f} blocks alive on average (256 CUs)")
names = ["loads + cell placement", "barrier 1", "counter scan (one wave) + barrier", "walk + emit to staging", "barrier 2", "run copy-out"]
print("phase means per level (us):  " + " | ".join(names) + " | items")
for lv in range(L):
    blk = fill[lv * U:(lv + 1) * U]
    ph = [(blk[:, i + 1] - blk[:, i]).mean() * TICK for i in range(6)]
    print(f"  level {lv:2d}: " + " ".join(f"{p:6.2f}" for p in ph) + f" | life {((blk[:, 6] - blk[:, 0]).mean() * TICK):6.2f} | items/unit {blk[:, 7].mean():7.0f}"
          f" | starts {(blk[:, 0].min() - t0) * TICK:6.1f}..{(blk[:, 0].max() - t0) * TICK:6.1f}")
ph = [(fill[:, i + 1] - fill[:, i]).mean() * TICK for i in range(6)]
print("  all      : " + " ".join(f"{p:6.2f}" for p in ph))
if fill[:, 12].min() >= t0:                                    # finer stamps inside the walk (thread 0: sample 0..3, final emit)
    for lv in (0, 4, 8, 12, 15):
        blk = fill[lv * U:(lv + 1) * U]
        seq = [3, 8, 9, 10, 11, 12, 4]
        print(f"  level {lv:2d} walk: " + " ".join(f"{(blk[:, b_] - blk[:, a_]).mean() * TICK:5.2f}" for a_, b_ in zip(seq[:-1], seq[1:]))
              + "   (to sample 0 | sample 0 | 1 | 2 | 3 | final emit)")

# ---- accumulate pass
offs = enc.offsets_host if hasattr(enc, "offsets_host") else enc.offsets.cpu().numpy()
offs = np.asarray(offs).astype(np.int64)
sizes = offs[1:] - offs[:-1]
P = (sizes + 4095) // 4096
BK = 64 if M >= 400000 else 32                                   # BK_TARGET of the launch (gridencoder.hip)
SUB = np.where(P >= BK, 1, (BK + P - 1) // np.maximum(P, 1))
first = np.concatenate([[0], np.cumsum(P * SUB)])
nacc = 512
acc = st[4096:4096 + nacc]
a0 = acc[:, 0].min()
rows = []
for b in range(nacc):
    for k in range(5):
        base = 1 + 6 * k
        tk = acc[b, base + 1]
        if acc[b, base + 5] < acc[b, 0] or acc[b, base] < acc[b, 0]:       # not written by the last launch
            break
        lv = int(np.searchsorted(first, tk, side="right") - 1)
        prev_end = acc[b, 0] if k == 0 else acc[b, base - 1]
        rows.append((lv, (acc[b, base] - prev_end) * TICK, (acc[b, base + 2] - acc[b, base]) * TICK, (acc[b, base + 3] - acc[b, base + 2]) * TICK,
                     (acc[b, base + 4] - acc[b, base + 3]) * TICK, (acc[b, base + 5] - acc[b, base + 4]) * TICK, (acc[b, base + 5] - a0) * TICK, k))
rows = np.array(rows)
print(f"\nACC: {len(rows)} stamped tasks (up to 5 per block); kernel span {rows[:, 6].max():.1f} us")
print("per level: tasks | ticket wait | old-value request + zero + barrier | items (thread 0) | barrier (slowest wave) | merge / flush | last end")
for lv in range(L):
    r = rows[rows[:, 0] == lv]
    if len(r):
        print(f"  level {lv:2d}: {len(r):4d} | " + " ".join(f"{r[:, i].mean():6.2f}" for i in range(1, 6)) + f" | {r[:, 6].max():6.1f} | task no. {r[:, 7].mean():.1f}")
print("  all      :      | " + " ".join(f"{rows[:, i].mean():6.2f}" for i in range(1, 6)))
per_blk = [(acc[b, 1:31].max() - acc[b, 0]) * TICK for b in range(nacc)]
print(f"block busy time: mean {np.mean(per_blk):.1f} us, max {np.max(per_blk):.1f}")
