"""Do the hash-grid encoder (L2-request-bound) and the fused head (MFMA / VALU-bound) overlap when launched on two streams?
The frame loop runs them back to back per iteration; if they overlap, an iteration's rows could be split in halves and head(A)
run beside encoder(B).  Times, for R tile-ordered frame rows: encoder alone, head alone, both on one stream, both on two streams."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from laenerf_amd import synthetic as S
from laenerf_amd.backend import gridencoder_backend as G, ffmlp_backend as R

dev = torch.device("cuda", 0)
net, r = bench.eval_model(dev, bound=2, seed=1234)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 21
# coherent rows: points along tile-ordered rays of the 1080p frame (8 samples per ray)
H, W = 1080, 1920
o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
idx, inv = r._tile_perm(H, W, 8, 8, dev)
n_rays = rows // 8
o, d = o[idx][:n_rays], d[idx][:n_rays]
t = 1.0 + 0.0035 * torch.arange(8, device=dev)[None, :, None]
xyz = (o[:, None] + d[:, None] * t).reshape(-1, 3).contiguous()
dirs = d[:, None].expand(-1, 8, -1).reshape(-1, 3).contiguous()
enc = net.encoder
table = enc.embeddings.detach().half().contiguous()
M = xyz.shape[0]
feats = [torch.empty(16, M, 2, device=dev, dtype=torch.half) for _ in range(2)]
h = torch.empty(M, 16, device=dev, dtype=torch.half); sig = torch.empty(M, device=dev); rgb = torch.empty(M, 3, device=dev)
sw, cw = net.sigma_net.weights.detach().half().contiguous(), net.color_net.weights.detach().half().contiguous()
S_ = float(np.log2(enc.per_level_scale))


def run_enc(k):
    G.grid_encode_forward(xyz, table, enc.offsets, feats[k], M, 3, 2, 16, S_, 16, None, 0, False, 0, in_map=(2.0, 0.25), offsets_host=enc.offsets_host)


def run_head(k):
    R.nerf_head_forward(feats[k], dirs, sw, cw, M, 1.0, h, sig, rgb, level_major=True)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


s2 = torch.cuda.Stream()
run_enc(0); run_enc(1); torch.cuda.synchronize()
t_enc, t_head = timed(lambda: run_enc(0)), timed(lambda: run_head(1))


def both_seq():
    run_enc(0); run_head(1)


def both_par():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        run_head(1)
    run_enc(0)
    torch.cuda.current_stream().wait_stream(s2)


print(f"rows {M}: encoder {t_enc:.1f} us, head {t_head:.1f} us, one stream {timed(both_seq):.1f} us, two streams {timed(both_par):.1f} us")
