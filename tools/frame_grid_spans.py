"""per-XCD busy spans of the encoder launch of ONE frame-loop iteration (library built with -DLAE_GRID_STAMPS; the frame is cut
short after LAE_FRAME_STOP_AFTER iterations so that the launch of interest is the last one): is the level -> XCD schedule, whose
cost table was calibrated on training batches (random pixels), balanced on a frame's ray-ordered rows?
gpurun -- 'LAE_BUILD_EXTRA_FLAGS=-DLAE_GRID_STAMPS python3 -m laenerf_amd.build --force && python3 tools/frame_grid_spans.py 3 10 30 60'"""
import ctypes, os, subprocess, sys
import numpy as np

if len(sys.argv) > 2:                                      # one child per iteration count (the knob is read once per process)
    for k in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, k], env=dict(os.environ, LAE_FRAME_STOP_AFTER=k))
    sys.exit(0)
import torch
sys.path.insert(0, os.getcwd())
from laenerf_amd import _lib, synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
dev = torch.device("cuda:0")
torch.manual_seed(0)
scene = os.environ.get("LAE_SPANS_SCENE", "800")
if scene == "800":
    net = NeRFNetwork(bound=1).to(dev).eval()
    r = NeRFRenderer(net, bound=1).to(dev).eval()
    r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
    H, W = 800, 800
    o, d = S.frame_rays(H, W)
else:
    import bench
    net, r = bench.eval_model(dev, bound=2, seed=1234)
    H, W = 1080, 1920
    o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
lib = _lib.load()
fn = lib.lae_debug_grid_stamps; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]; fn.restype = ctypes.c_int
for rep in range(3):
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        r.render_eval(o, d, bg_color=1, max_steps=1024, image_hw=(H, W))
    torch.cuda.synchronize()
buf = np.zeros((32768 * 8, 4), dtype=np.uint64)
assert fn(buf.ctypes.data, buf.nbytes) == 0
st = buf.astype(np.int64)
ok = st[:, 1] > 0
# the last launch only: slots of blocks without rows in it still hold an earlier iteration's stamps; between two launches no
# encoder block starts for tens of microseconds (head + emit kernels), inside one they start back to back
starts = np.sort(st[ok, 0])
gaps = np.nonzero(np.diff(starts) > 2000)[0]                # 20 us of the 100 MHz clock
if len(gaps):
    ok &= st[:, 0] >= starts[gaps[-1] + 1]
idx = np.nonzero(ok)[0]
t0 = st[idx, 0].min()
xcd = idx & 7
lv = st[idx, 2]
print(f"scene {scene}, after {os.environ.get('LAE_FRAME_STOP_AFTER')} iterations: {len(idx)} stamped blocks; span {(st[idx, 1].max() - t0) * 0.01:.1f} us; XCD ends (us): "
      + " ".join(f"{(st[idx[xcd == x], 1].max() - t0) * 0.01:5.1f}" for x in range(8)))
print("   level [first start..last end] us: " + " ".join(f"{l}:[{(st[idx[lv == l], 0].min() - t0) * 0.01:.0f}..{(st[idx[lv == l], 1].max() - t0) * 0.01:.0f}]" for l in range(16) if (lv == l).any()))
print("   level on XCD: " + " ".join(f"{l}:{sorted(set((idx[lv == l] & 7).tolist()))}" for l in range(16) if (lv == l).any()))
