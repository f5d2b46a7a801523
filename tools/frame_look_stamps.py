"""where the waves of ONE k_frame_lookahead launch spend their time (needs -DLAE_FRAME_STAMPS, see tools/grid_bwd_stamps.py):
    LAE_BUILD_EXTRA_FLAGS=-DLAE_FRAME_STAMPS python -m laenerf_amd.build --force; gpurun -- python tools/frame_look_stamps.py [iteration ...]
LAE_STAMPS_SCENE=whole|shard: the configs[3]-shaped 1080p frame / rank 0's shard of 8 instead of the 800x800 lego-shaped frame"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import _lib, synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
dev = torch.device("cuda:0")
scene = os.environ.get("LAE_STAMPS_SCENE", "800")
budget = 0
if scene == "800":
    torch.manual_seed(0)
    net = NeRFNetwork(bound=1).to(dev).eval()
    r = NeRFRenderer(net, bound=1).to(dev).eval()
    r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
    o, d = S.frame_rays(800, 800)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
else:
    import bench
    from laenerf_amd.dist import shard_indices_device
    net, r = bench.eval_model(dev, bound=2, seed=1234)
    o, d = S.frame_rays(1080, 1920, focal=1111.1 * 1080 / 800, radius=1.6)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    if scene == "shard":
        idx = shard_indices_device(o.shape[0], 0, 8, dev)
        o, d = o[idx].contiguous(), d[idx].contiguous()
        budget = 1080 * 1920
lib = _lib.load()
fn = lib.lae_debug_look_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]; fn.restype = ctypes.c_int
for phase in ([int(a) for a in sys.argv[1:]] or [5, 30, 60]):
    assert fn(None, 0, phase) == 0
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            r.render_eval(o, d, bg_color=1, max_steps=1024, row_budget=budget)
    torch.cuda.synchronize()
    buf = np.zeros((16384, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.nbytes, -1) == 0
    st = buf.astype(np.int64)
    ok = st[:, 3] > 0
    w = st[ok]
    t0 = w[:, 0].min()
    life = (w[:, 3] - w[:, 0]) * 0.01
    rounds, coop, rays = w[:, 4] & 0xffff, (w[:, 4] >> 16) & 0xffff, w[:, 4] >> 32
    print(f"iteration {phase}: {ok.sum()} waves with rays; kernel span {(w[:, 3].max() - t0) * 0.01:.1f} us; last wave STARTS at {(w[:, 0].max() - t0) * 0.01:.1f} us")
    print(f"  wave life: mean {life.mean():.1f} us, p50 {np.median(life):.1f}, p90 {np.percentile(life, 90):.1f}, max {life.max():.1f}; state loads {((w[:, 1] - w[:, 0]) * 0.01).mean():.1f} us")
    print(f"  lane rounds per wave: mean {rounds.mean():.2f}, max {rounds.max()}; waves that went cooperative {np.mean(coop > 0) * 100:.0f} %, rays finished cooperatively per such wave {coop[coop > 0].mean() if (coop > 0).any() else 0:.1f}; rays per wave {rays.mean():.1f}")
    c = coop > 0
    if c.any():
        print(f"  cooperative phase of those waves: mean {((w[c, 3] - w[c, 2]) * 0.01).mean():.1f} us, max {((w[c, 3] - w[c, 2]) * 0.01).max():.1f}; lane phase before it {((w[c, 2] - w[c, 1]) * 0.01).mean():.1f} us")
    slow = np.argsort(life)[-5:]
    for i in slow:
        print(f"    slow wave: life {life[i]:.1f} us, start {(w[i, 0] - t0) * 0.01:.1f}, loads {(w[i, 1] - w[i, 0]) * 0.01:.1f}, rounds {rounds[i]}, coop rays {coop[i]}, coop phase {(w[i, 3] - w[i, 2]) * 0.01 if coop[i] else 0:.1f} us")
