#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference's Python (unmodified) in THIS container.

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (never available on the GPU box; only the emitted vectors travel).

What is executed from the reference: gridencoder/grid.py (GridEncoder sizing), ffmlp/ffmlp.py (FFMLP
parameter layout/init, padding), nerf/network.py (the nn.Linear chain + trunc_exp + sigmoid that ffmlp
replaces), nerf/network_ff.py and nerf/renderer.py (run / run_cuda / run_cuda_distill orchestration),
raymarching/raymarching.py, shencoder/sphere_harmonics.py, activation.py, encoding.py.
What is NOT from the reference: the four native extensions.  The reference's CUDA kernels cannot be
built here, so `_raymarching/_gridencoder/_shencoder/_ffmlp` are the repo's CPU oracle
(tests/golden/oracle_backends.py).  The vectors therefore pin (a) everything the reference computes in
Python and (b) the behaviour of this repo's operators UNDER the reference's own callers.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("LAE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

sys.path.insert(0, HERE)
import oracle_backends  # noqa: E402

oracle_backends.install()

# ---- stubs for modules the reference imports but the hot path never uses
sys.modules["trimesh"] = types.ModuleType("trimesh")
sys.modules["turtle"] = types.ModuleType("turtle")
sys.modules["turtle"].backward = sys.modules["turtle"].forward = None
sys.path.insert(0, REF)
nerf_pkg = types.ModuleType("nerf")
nerf_pkg.__path__ = [os.path.join(REF, "nerf")]
sys.modules["nerf"] = nerf_pkg
nerf_utils = types.ModuleType("nerf.utils")
nerf_utils.custom_meshgrid = lambda *a: torch.meshgrid(*a, indexing="ij")     # nerf/utils.py:43-48
sys.modules["nerf.utils"] = nerf_utils
torch.Tensor.cuda = lambda self, *a, **k: self          # wrappers call .cuda() unconditionally

# the fused-MLP oracle speaks fp16; without a GPU autocast cannot cast for us, so adapt fp32 tensors that
# carry fp16-representable values (the reference would hand over half tensors under `-O`)
_ff = sys.modules["_ffmlp"]
_fwd, _inf, _bwd = _ff.ffmlp_forward, _ff.ffmlp_inference, _ff.ffmlp_backward


def _h(t):
    return t.to(torch.float16).contiguous()


def _ff_forward(inputs, weights, B, i, o, h, n, a, oa, forward_buffer, outputs):
    fb, out = torch.empty(forward_buffer.shape, dtype=torch.float16), torch.empty(outputs.shape, dtype=torch.float16)
    _fwd(_h(inputs), _h(weights), B, i, o, h, n, a, oa, fb, out)
    forward_buffer.copy_(fb); outputs.copy_(out)


def _ff_inference(inputs, weights, B, i, o, h, n, a, oa, inference_buffer, outputs):
    out = torch.empty(outputs.shape, dtype=torch.float16)
    _inf(_h(inputs), _h(weights), B, i, o, h, n, a, oa, None, out)
    outputs.copy_(out)


def _ff_backward(grad, inputs, weights, forward_buffer, B, i, o, h, n, a, oa, calc_gi, backward_buffer, grad_inputs, grad_weights):
    bb = torch.empty(backward_buffer.shape, dtype=torch.float16)
    gw = torch.empty(grad_weights.shape, dtype=torch.float16)
    gi = torch.empty(inputs.shape, dtype=torch.float16) if calc_gi else None
    _bwd(_h(grad), _h(inputs), _h(weights), _h(forward_buffer), B, i, o, h, n, a, oa, calc_gi, bb, gi, gw)
    backward_buffer.copy_(bb); grad_weights.copy_(gw)
    if calc_gi:
        grad_inputs.copy_(gi)


_ff.ffmlp_forward, _ff.ffmlp_inference, _ff.ffmlp_backward = _ff_forward, _ff_inference, _ff_backward

from gridencoder import GridEncoder  # noqa: E402  (reference)
from ffmlp import FFMLP  # noqa: E402  (reference)
import raymarching  # noqa: E402  (reference wrappers)
from nerf.network import NeRFNetwork as LinearNet  # noqa: E402
from nerf.network_ff import NeRFNetwork as FFNet  # noqa: E402
from nerf.renderer import NeRFRenderer  # noqa: E402
import nerf.renderer as ref_renderer  # noqa: E402

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from laenerf_amd import synthetic as S  # noqa: E402  (numpy-only generators)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def f16r(t):
    """round to fp16-representable fp32"""
    return t.to(torch.float16).to(torch.float32)


# ---------------------------------------------------------------- A. GridEncoder level sizing (grid.py:98-134)
def gen_grid_offsets():
    cfgs = [dict(), dict(desired_resolution=2048), dict(desired_resolution=4096), dict(num_levels=4, desired_resolution=2048),
            dict(input_dim=2, num_levels=4, log2_hashmap_size=19, desired_resolution=2048),
            dict(desired_resolution=2048, align_corners=True), dict(desired_resolution=1024, gridtype="tiled", log2_hashmap_size=15),
            dict(num_levels=8, level_dim=4, base_resolution=8, per_level_scale=1.5, log2_hashmap_size=14)]
    out = {}
    for i, kw in enumerate(cfgs):
        e = GridEncoder(**kw)
        out[f"cfg{i}_offsets"] = e.offsets.numpy()
        out[f"cfg{i}_pls"] = np.float64(e.per_level_scale)
        out[f"cfg{i}_kw"] = np.array(repr(sorted(kw.items())))
    save("grid_offsets", **out)


# ---------------------------------------------------------------- B. FFMLP layout / init (ffmlp.py:99-165)
def gen_ffmlp_init():
    out = {}
    for name, args in (("sigma", (32, 16, 64, 2)), ("color", (32, 3, 64, 3)), ("wide", (48, 3, 128, 2))):
        m = FFMLP(*args)
        w = m.weights.detach().numpy()
        out[name + "_n"] = np.int64(m.num_parameters)
        out[name + "_head"] = w[:256].copy()
        out[name + "_sum"] = np.float64(w.astype(np.float64).sum())
        out[name + "_padded_out"] = np.int64(m.padded_output_dim)
    # padding rule of FFMLP.forward (ffmlp.py:157-165), observed through the output shape of the unpad
    m = FFMLP(32, 3, 64, 2).eval()
    seen = []

    def spy(inputs, weights, B, *a):
        seen.append(B)
        a[-1].zero_()
    _ff.ffmlp_inference, keep = spy, _ff.ffmlp_inference
    import ffmlp.ffmlp as ffmod
    for B in (1, 127, 128, 129, 256):
        y = m(torch.zeros(B, 32))
        assert y.shape == (B, 3)
    _ff.ffmlp_inference = keep
    out["pad_B_in"] = np.array([1, 127, 128, 129, 256])
    out["pad_B_backend"] = np.array(seen)
    save("ffmlp_init", **out)


def small_encoder(num_levels=16, log2_hashmap_size=10, desired_resolution=2048, seed=1, scale=0.5):
    e = GridEncoder(num_levels=num_levels, log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution)
    g = torch.Generator().manual_seed(seed)
    e.embeddings.data = f16r((torch.rand(e.embeddings.shape, generator=g) * 2 - 1) * scale)
    return e


# ---------------------------------------------------------------- C. nn.Linear chain (network.py:95-124) = oracle for ffmlp
def gen_mlp_chain():
    torch.manual_seed(7)
    net = LinearNet(num_layers=3, hidden_dim=64, geo_feat_dim=15, num_layers_color=4, hidden_dim_color=64, bound=1,
                    cuda_ray=False)
    net.encoder = small_encoder()
    for p in list(net.sigma_net.parameters()) + list(net.color_net.parameters()):
        p.data = f16r(p.data)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(512, 3, generator=g) * 2 - 1) * 0.9
    d = torch.randn(512, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        sigma, color = net(x, d)
        enc = net.encoder(x, bound=1)
        h = enc
        hs = []
        for l in range(3):
            h = net.sigma_net[l](h)
            if l != 2:
                h = torch.relu(h)
            hs.append(h.clone())
    save("mlp_chain", x=x.numpy(), d=d.numpy(), table=net.encoder.embeddings.detach().numpy(),
         offsets=net.encoder.offsets.numpy(), pls=np.float64(net.encoder.per_level_scale), enc=enc.numpy(),
         sigma_w=[p.detach().numpy() for p in net.sigma_net.parameters()][0], sigma_w1=net.sigma_net[1].weight.detach().numpy(),
         sigma_w2=net.sigma_net[2].weight.detach().numpy(),
         color_w0=net.color_net[0].weight.detach().numpy(), color_w1=net.color_net[1].weight.detach().numpy(),
         color_w2=net.color_net[2].weight.detach().numpy(), color_w3=net.color_net[3].weight.detach().numpy(),
         sigma_h=hs[2].numpy(), sigma=sigma.numpy(), color=color.numpy())


# ---------------------------------------------------------------- D. NeRFRenderer.run cumprod compositing (renderer.py:128-256)
class AnalyticField(NeRFRenderer):
    """analytic density/colour so the compositing of `run` can be checked on known samples"""

    def density(self, x):
        r2 = (x * x).sum(-1)
        return {"sigma": 40.0 * torch.exp(-6.0 * r2)}

    def color(self, x, d, mask=None, **kw):
        rgb = torch.sigmoid(torch.stack([3 * x[:, 0] + d[:, 1], 2 * x[:, 1] - d[:, 2], x[:, 2] * 4 + d[:, 0]], -1))
        if mask is not None:
            rgb = rgb * mask[:, None]
        return rgb


def gen_run_path():
    o, d = S.lego_like_rays(192, seed=5)
    r = AnalyticField(bound=1, cuda_ray=False, min_near=0.2).eval()
    with torch.no_grad():
        res = r.run(torch.from_numpy(o)[None], torch.from_numpy(d)[None], num_steps=96, upsample_steps=0, bg_color=1, perturb=False)
    save("run_path", rays_o=o, rays_d=d, image=res["image"][0].numpy(), depth=res["depth"][0].numpy(),
         weights_sum=res["weights_sum"].numpy(), num_steps=np.int64(96))


# ---------------------------------------------------------------- D2. BASELINE configs[0]: the `run` path end to end
def gen_cfg0():
    """configs[0] (lego 64x64, 1024 rays/batch, L=4 hash grid, nn.Linear nets, cuda_ray off): ONE train step of the
    reference's own NeRFNetwork (nerf/network.py) through NeRFRenderer.run (renderer.py:128-256) with the settings of
    main_nerf.py:32-35 (num_steps 512, upsample_steps 0), fp32, loss = MSE, gradients of every parameter.
    The table is capped at 2^14 entries per level so that the fixture stays small (BASELINE's T only changes sizes)."""
    torch.manual_seed(5)
    net = LinearNet(bound=1, cuda_ray=False, min_near=0.2)
    enc = GridEncoder(num_levels=4, log2_hashmap_size=14, desired_resolution=2048)
    g = torch.Generator().manual_seed(2)
    enc.embeddings.data = f16r((torch.rand(enc.embeddings.shape, generator=g) * 2 - 1) * 0.5)
    net.encoder, net.in_dim = enc, enc.output_dim
    net.sigma_net[0] = torch.nn.Linear(enc.output_dim, net.hidden_dim, bias=False)
    for p_ in list(net.sigma_net.parameters()) + list(net.color_net.parameters()):
        p_.data = f16r(p_.data)
    o, d = S.lego_like_rays(1024, H=64, W=64, focal=1111.1 * 64 / 800, seed=3)
    target = torch.rand(1024, 3, generator=torch.Generator().manual_seed(8))
    net.train()
    res = net.run(torch.from_numpy(o)[None], torch.from_numpy(d)[None], num_steps=512, upsample_steps=0, bg_color=1, perturb=False)
    loss = ((res["image"][0] - target) ** 2).mean()
    loss.backward()
    out = dict(rays_o=o, rays_d=d, target=target.numpy(), table=enc.embeddings.detach().numpy().astype(np.float16),
               offsets=enc.offsets.numpy(), pls=np.float64(enc.per_level_scale),
               image=res["image"][0].detach().numpy(), depth=res["depth"][0].detach().numpy(),
               weights_sum=res["weights_sum"].detach().numpy(), loss=np.float64(loss.item()),
               g_table_norm=np.float64(enc.embeddings.grad.norm().item()), g_table_sample=enc.embeddings.grad.numpy()[::211].copy(),
               num_steps=np.int64(512))
    for i, layer in enumerate(net.sigma_net):
        out[f"sigma_w{i}"], out[f"g_sigma_w{i}"] = layer.weight.detach().numpy(), layer.weight.grad.numpy().copy()
    for i, layer in enumerate(net.color_net):
        out[f"color_w{i}"], out[f"g_color_w{i}"] = layer.weight.detach().numpy(), layer.weight.grad.numpy().copy()
    save("cfg0_run_step", **out)


# ---------------------------------------------------------------- E. run_cuda / run_cuda_distill orchestration
def make_ff_net(bound, seed):
    torch.manual_seed(seed)
    net = FFNet(bound=bound, cuda_ray=True, min_near=0.2, density_thresh=10)
    net.encoder = small_encoder(desired_resolution=2048 * bound, seed=seed)
    net.sigma_net.weights.data = f16r(net.sigma_net.weights.data)
    net.color_net.weights.data = f16r(net.color_net.weights.data)
    return net


class half_table_path:
    """the reference's `-O` precision path for the hash grid, on the CPU: under cuda autocast grid.py:41-44 casts the table to
    half, the encoder accumulates and returns halves, and the backward receives a half gradient and adds halves into a half
    gradient table (gridencoder.cu:325-331).  Without a GPU autocast is off, so the backend registered as `_gridencoder` does
    those casts itself for the duration of this block (round 3: the GPU runs this very path, so the golden gradients can be
    asserted at rounding level instead of 5-8 %)."""

    def __enter__(self):
        ge = sys.modules["_gridencoder"]
        self.ge, self.fwd, self.bwd = ge, ge.grid_encode_forward, ge.grid_encode_backward
        fwd, bwd = self.fwd, self.bwd

        def forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
            assert dy_dx is None
            out_h = torch.empty(outputs.shape, dtype=torch.float16)
            fwd(inputs, embeddings.to(torch.float16).contiguous(), offsets, out_h, B, D, C, L, S, H, None, gridtype, align_corners, interp)
            outputs.copy_(out_h)

        def backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
            assert dy_dx is None and grad_inputs is None
            ge_h = torch.zeros(grad_embeddings.shape, dtype=torch.float16)
            bwd(grad.to(torch.float16).contiguous(), inputs, embeddings.to(torch.float16).contiguous(), offsets, ge_h, B, D, C, L, S, H, None,
                None, gridtype, align_corners, interp)
            grad_embeddings.add_(ge_h.float())
        ge.grid_encode_forward, ge.grid_encode_backward = forward, backward
        return self

    def __exit__(self, *a):
        self.ge.grid_encode_forward, self.ge.grid_encode_backward = self.fwd, self.bwd


E2E_LOSS_SCALE = 1024.0      # the fp16 gradients need the loss scale the reference trains with (GradScaler); the stored gradients are unscaled


def gen_e2e(tag, bound):
    with half_table_path():
        _gen_e2e(tag, bound)


def _gen_e2e(tag, bound):
    C = 1 + int(np.ceil(np.log2(bound)))
    net = make_ff_net(bound, seed=11)
    grid = S.sphere_density_grid(cascade=C, bound=float(bound), radius=0.55)
    bitfield = torch.from_numpy(S.pack_bits_np(grid, 10.0))
    net.density_bitfield = bitfield.clone()
    o, d = S.lego_like_rays(256, seed=9, radius=3.2 if bound == 1 else 2.5)
    ro, rd = torch.from_numpy(o)[None], torch.from_numpy(d)[None]
    gen = torch.Generator().manual_seed(21)
    target = torch.rand(256, 3, generator=gen)
    out = dict(rays_o=o, rays_d=d, bitfield=bitfield.numpy(), table=net.encoder.embeddings.detach().numpy(),
               offsets=net.encoder.offsets.numpy(), pls=np.float64(net.encoder.per_level_scale),
               sigma_w=net.sigma_net.weights.detach().numpy(), color_w=net.color_net.weights.detach().numpy(),
               target=target.numpy(), bound=np.float64(bound))
    # --- training render, first-steps mode (mean_count = 0 -> M = N*max_steps, trimmed)
    net.train()
    res = net.render(ro, rd, staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=256)
    loss = ((res["image"][0] - target) ** 2).mean()
    (loss * E2E_LOSS_SCALE).backward()
    gt = net.encoder.embeddings.grad / E2E_LOSS_SCALE
    out.update(train_image=res["image"][0].detach().numpy(), train_depth=res["depth"][0].detach().numpy(),
               train_ws=res["weights_sum"].detach().numpy(), train_counter=net.step_counter[0].numpy().copy(),
               train_loss=np.float64(loss.item()), loss_scale=np.float64(E2E_LOSS_SCALE),
               g_sigma_w=(net.sigma_net.weights.grad / E2E_LOSS_SCALE).numpy().copy(),
               g_color_w=(net.color_net.weights.grad / E2E_LOSS_SCALE).numpy().copy(),
               g_table_norm=np.float64(gt.norm().item()), g_table_sample=gt.numpy()[::997].copy(),
               g_table_nonzero=np.int64((gt != 0).sum().item()))
    # --- steady-state mode: mean_count > 0 (M = mean_count rounded up, may drop rays that overflow)
    net.zero_grad()
    net.mean_count = int(net.step_counter[0, 0].item() * 0.6)        # deliberately too small: exercises the overflow drop
    res2 = net.render(ro, rd, staged=False, bg_color=1, perturb=False, force_all_rays=False, dt_gamma=0, max_steps=256)
    out.update(train2_mean_count=np.int64(net.mean_count), train2_image=res2["image"][0].detach().numpy(),
               train2_ws=res2["weights_sum"].detach().numpy())
    # --- eval render (renderer.py:335-387)
    net.eval()
    with torch.no_grad():
        ev = net.render(ro, rd, staged=True, bg_color=1, perturb=False, dt_gamma=0, max_steps=256, scale_depth=True)
    out.update(eval_image=ev["image"][0].numpy(), eval_depth=ev["depth"][0].numpy())
    # --- distill render (renderer.py:394-480) with an edit bitfield = one octant of the density bitfield
    edit_grid = grid.copy()
    cx, cy, cz = S._morton_inverse_table(128)
    edit_grid[:, ~((cx >= 64) & (cy >= 48))] = 0
    edit_bits = torch.from_numpy(S.pack_bits_np(edit_grid, 10.0))
    with torch.no_grad():
        ds = net.run_cuda_distill(ro, rd, edit_bits, dt_gamma=0, perturb=False, max_steps=256)
    out.update(edit_bitfield=edit_bits.numpy(), dist_image=ds["image"][0].numpy(), dist_depth=ds["depth"][0].numpy(),
               dist_depth_edit=ds["depth_edit"].numpy(), dist_weights_edit=ds["weights_edit"].numpy(),
               dist_weights=ds["weights"].numpy(), dist_x_term=ds["x_term"].numpy())
    save("e2e_" + tag, **out)


def gen_run_upsample():
    """`run` with importance resampling (sample_pdf, det=True in eval mode), renderer.py:170-207"""
    o, d = S.lego_like_rays(96, seed=6)
    r = AnalyticField(bound=1, cuda_ray=False, min_near=0.2).eval()
    with torch.no_grad():
        res = r.run(torch.from_numpy(o)[None], torch.from_numpy(d)[None], num_steps=48, upsample_steps=32, bg_color=1, perturb=False)
    g = torch.Generator().manual_seed(4)
    bins = torch.sort(torch.rand(7, 20, generator=g), dim=-1)[0]
    w = torch.rand(7, 19, generator=g) ** 3
    save("run_upsample", rays_o=o, rays_d=d, image=res["image"][0].numpy(), depth=res["depth"][0].numpy(),
         weights_sum=res["weights_sum"].numpy(), num_steps=np.int64(48), upsample_steps=np.int64(32),
         pdf_bins=bins.numpy(), pdf_weights=w.numpy(), pdf_samples=ref_renderer.sample_pdf(bins, w, 16, det=True).numpy())


# ---------------------------------------------------------------- E2. occupancy-grid maintenance (renderer.py:482-649)
def look_at_poses(n, radius, seed):
    """camera-to-world [n,4,4]: cameras on a sphere looking at the origin, +z forward (the convention of
    mark_untrained_grid: a point is in front when its camera-space z is positive)"""
    rng = np.random.default_rng(seed)
    P = np.zeros((n, 4, 4), np.float32)
    for i in range(n):
        p = rng.standard_normal(3); p = p / np.linalg.norm(p) * radius
        f = -p / np.linalg.norm(p)
        up = np.array([0, 0, 1.0]) if abs(f[2]) < 0.9 else np.array([0, 1.0, 0])
        r = np.cross(up, f); r /= np.linalg.norm(r)
        u = np.cross(f, r)
        P[i, :3, 0], P[i, :3, 1], P[i, :3, 2], P[i, :3, 3], P[i, 3, 3] = r, u, f, p, 1
    return P


def reference_get_rays():
    """nerf/utils.py cannot be imported here (tensorboardX, cv2, lpips, tinycudann ... are absent), and only two of its
    functions are needed: `custom_meshgrid` (:43-48) and `get_rays` (:60-153).  They are compiled from the reference file
    where it lies (nothing is copied) into a namespace that holds what the module itself would have imported for them."""
    import ast
    path = os.path.join(REF, "nerf", "utils.py")
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("custom_meshgrid", "get_rays")]
    assert len(keep) == 2
    from packaging import version as pver
    ns = {"torch": torch, "pver": pver}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["get_rays"]


def gen_get_rays():
    get_rays = reference_get_rays()
    out = {}
    poses = torch.from_numpy(look_at_poses(3, 3.2, seed=5))
    for tag, kw in (("all", dict(B=1, H=6, W=5, N=-1, intr=(7.5, 7.0, 2.4, 3.1))),
                    ("rand", dict(B=2, H=37, W=53, N=64, intr=(60.0, 61.5, 26.5, 18.5))),
                    ("perturb", dict(B=1, H=9, W=7, N=-1, intr=(11.0, 11.0, 3.5, 4.5), perturb_ray_dirs=True)),
                    ("patch", dict(B=1, H=40, W=48, N=64, intr=(50.0, 50.0, 24.0, 20.0), patch_size=4)),
                    ("emap", dict(B=2, H=300, W=200, N=96, intr=(250.0, 250.0, 100.0, 150.0), error_map=True)),
                    ("lego", dict(B=1, H=800, W=800, N=4096, intr=(1111.111, 1111.111, 400.0, 400.0)))):
        B, H, W, N, intr = kw["B"], kw["H"], kw["W"], kw["N"], kw["intr"]
        torch.manual_seed(17)
        emap = torch.rand(B, 128 * 128) + 0.01 if kw.get("error_map") else None
        torch.manual_seed(23)
        res = get_rays(poses[:B], torch.tensor(intr).numpy(), H, W, N, error_map=emap, patch_size=kw.get("patch_size", 1),
                       perturb_ray_dirs=kw.get("perturb_ray_dirs", False))
        out[f"{tag}_poses"] = poses[:B].numpy()
        out[f"{tag}_cfg"] = np.array([H, W, N], np.int64)
        out[f"{tag}_intr"] = np.array(intr, np.float32)
        out[f"{tag}_rays_o"] = res["rays_o"].numpy()
        out[f"{tag}_rays_d"] = res["rays_d"].numpy()
        if "inds" in res:
            out[f"{tag}_inds"] = res["inds"].contiguous().numpy()
        if "inds_coarse" in res:
            out[f"{tag}_inds_coarse"] = res["inds_coarse"].numpy()
        if kw.get("perturb_ray_dirs"):
            torch.manual_seed(23)                     # the only draw of this case: offset = rand(2) - 0.5 (:133)
            out[f"{tag}_offset"] = (torch.rand(2) - 0.5).numpy()
    save("get_rays", **out)


def gen_density_grid():
    bound, H = 2, 16
    C = 2
    net = make_ff_net(bound, seed=23)
    net.grid_size = H
    net.density_grid = torch.zeros(C, H ** 3)
    net.density_bitfield = torch.zeros(C * H ** 3 // 8, dtype=torch.uint8)
    net.train()
    out = dict(table=net.encoder.embeddings.detach().numpy(), offsets=net.encoder.offsets.numpy(),
               pls=np.float64(net.encoder.per_level_scale), sigma_w=net.sigma_net.weights.detach().numpy(),
               color_w=net.color_net.weights.detach().numpy(), bound=np.float64(bound), H=np.int64(H))
    poses = look_at_poses(5, 1.3, seed=2)
    intr = np.array([40.0, 40.0, 16.0, 16.0])
    net.mark_untrained_grid(poses, intr)
    out.update(poses=poses, intrinsics=intr, grid_marked=net.density_grid.numpy().copy())

    rec = {"rand": [], "randint": [], "sigma": []}
    real_rand_like, real_randint, real_density = torch.rand_like, torch.randint, net.density

    def rand_like(t, **kw):
        r = real_rand_like(t, **kw); rec["rand"].append(r.numpy().copy()); return r

    def randint(*a, **kw):
        r = real_randint(*a, **kw); rec["randint"].append(r.numpy().copy()); return r

    def density(x):
        o = real_density(x); rec["sigma"].append(o["sigma"].detach().numpy().copy()); return o

    torch.manual_seed(31)
    torch.rand_like, torch.randint, net.density = rand_like, randint, density
    try:
        net.update_extra_state()                                   # iter_density 0: full sweep
        for c in range(C):
            out[f"full_noise{c}"], out[f"full_sigma{c}"] = rec["rand"][c], rec["sigma"][c]
        out.update(full_grid=net.density_grid.numpy().copy(), full_bitfield=net.density_bitfield.numpy().copy(),
                   full_mean=np.float64(net.mean_density))
        rec["rand"].clear(); rec["randint"].clear(); rec["sigma"].clear()
        net.iter_density = 16                                      # partial sweep
        net.local_step = 3
        net.step_counter[:3, 0] = torch.tensor([1000, 1200, 1100], dtype=torch.int32)
        net.update_extra_state()
        for c in range(C):
            out[f"part_coords{c}"], out[f"part_pick{c}"] = rec["randint"][2 * c], rec["randint"][2 * c + 1]
            out[f"part_noise{c}"], out[f"part_sigma{c}"] = rec["rand"][c], rec["sigma"][c]
        out.update(part_grid=net.density_grid.numpy().copy(), part_bitfield=net.density_bitfield.numpy().copy(),
                   part_mean=np.float64(net.mean_density), part_mean_count=np.int64(net.mean_count))
    finally:
        torch.rand_like, torch.randint, net.density = real_rand_like, real_randint, real_density
    save("density_grid", **out)


# ---------------------------------------------------------------- F. operator-level vectors under the reference's wrappers
def gen_ops():
    """small vectors produced THROUGH the reference's Python wrappers (shape/padding/alignment rules of
    raymarching.py:161-235, 297-348 are reference code; the arithmetic is the oracle's)."""
    o, d = S.lego_like_rays(128, seed=13)
    ro, rd = torch.from_numpy(o), torch.from_numpy(d)
    grid = S.sphere_density_grid()
    bits = torch.from_numpy(S.pack_bits_np(grid, 10.0))
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0])
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, counter, -1, False, 128,
                                                            False, 0, 128)
    out = dict(rays_o=o, rays_d=d, bitfield=bits.numpy(), nears=nears.numpy(), fars=fars.numpy(), counter=counter.numpy(),
               xyzs=xyzs.numpy(), deltas=deltas.numpy(), rays=rays.numpy(), M_trimmed=np.int64(xyzs.shape[0]))
    # mean_count path: M = mean_count rounded up by `+= align - m % align`
    for mc in (1000, 1024, 5000):
        c2 = torch.zeros(2, dtype=torch.int32)
        x2, _, _, r2 = raymarching.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, c2, mc, False, 128, False, 0, 128)
        out[f"mc{mc}_M"] = np.int64(x2.shape[0])
        out[f"mc{mc}_rays"] = r2.numpy()
    # inference march output sizing (align 128)
    alive = torch.arange(100, dtype=torch.int32)
    t = nears.clone()
    x3, _, dl3 = raymarching.march_rays(100, 3, alive, t, ro, rd, 1.0, bits, 1, 128, nears, fars, 128, False, 0, 128)
    out["infer_M"] = np.int64(x3.shape[0])
    out["infer_xyzs"] = x3.numpy()
    out["infer_deltas"] = dl3.numpy()
    save("ops_wrappers", **out)


# ---------------------------------------------------------------- K. EditGrid region growing (editing/editgrid.py:80-136, 274-340)
def gen_editgrid():
    """the reference's EditGrid.new_from_points + grow_region_queue executed here on CPU tensors (its `_edit_grid`
    extension is not used by these two methods and is stubbed; morton3D comes from the oracle backend)"""
    from types import SimpleNamespace
    sys.modules.setdefault("_edit_grid", types.ModuleType("_edit_grid"))
    from editing.editgrid import EditGrid
    H, V = 128, 128 ** 3
    cases = {}
    for tag, cascade, bound, seeds, iters in (("c1", 1, 1.0, [[0.02, -0.03, 0.05]], 1500),
                                              ("c2", 2, 2.0, [[0.4, 0.1, -0.2], [-1.3, 0.2, 0.4]], 900)):
        rng = np.random.default_rng(len(tag) + cascade)
        # density: a few blobs per cascade, in Morton order like density_grid
        c = np.stack(np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
        from oracle import oracle as O
        midx = O.morton3D(c)
        dens = np.zeros((cascade, V), np.float32)
        for cas in range(cascade):
            b = min(2.0 ** cas, bound)
            xyz = (2 * (c + 0.5) / H - 1) * b
            field = np.zeros(V, np.float32)
            for centre, rad in (((0.0, 0.0, 0.0), 0.18), ((0.4, 0.1, -0.2), 0.15), ((-1.3, 0.2, 0.4), 0.3)):
                field = np.maximum(field, np.where(np.linalg.norm(xyz - np.array(centre), axis=1) < rad, 15.0 + rng.random(V).astype(np.float32) * 10, 0))
            dens[cas, midx] = field
        trainer = SimpleNamespace(model=SimpleNamespace(density_bitfield=torch.zeros(cascade * V // 8, dtype=torch.uint8), cascade=cascade))
        eg = EditGrid()
        pts = torch.tensor(seeds, dtype=torch.float32)
        eg.new_from_points(pts, trainer=trainer, bound=bound)
        grid0 = eg.grid.numpy().copy()
        q0 = np.array([[int(v) for v in cc.tolist()] + [int(l)] for cc, l in eg.growing_queue], np.int32)
        eg.grow_region_queue(torch.from_numpy(dens), 12.0, grow_iterations=iters)
        q1 = np.array([[int(v) for v in cc.tolist()] + [int(l)] for cc, l in eg.growing_queue], np.int32).reshape(-1, 4)
        nz = np.nonzero(dens.reshape(-1))[0]
        cases.update({f"{tag}_pts": np.array(seeds, np.float32), f"{tag}_cascade": cascade, f"{tag}_bound": bound, f"{tag}_iters": iters,
                      f"{tag}_dens_idx": nz.astype(np.int32), f"{tag}_dens_val": dens.reshape(-1)[nz],
                      f"{tag}_grid0": np.nonzero(np.unpackbits(grid0, bitorder="little"))[0].astype(np.int32),
                      f"{tag}_queue0": q0, f"{tag}_grid1": np.nonzero(np.unpackbits(eg.grid.numpy(), bitorder="little"))[0].astype(np.int32),
                      f"{tag}_queue1": q1})
        print(tag, "selected", cases[f"{tag}_grid1"].size, "queue", q0.shape[0], "->", q1.shape[0])
    save("editgrid", **cases)


def gen_freq():
    """K18: the reference ships a pure-torch frequency encoder (encoding.py:5-43; `get_encoder('frequency')` builds the CUDA
    one, whose row layout -- inputs, then per frequency the sines followed by the cosines -- it shares): executed here"""
    from encoding import FreqEncoder as RefFreq
    gen = torch.Generator().manual_seed(3)
    out = {}
    for D, deg in ((3, 4), (3, 10), (2, 6), (5, 1)):
        x = (torch.rand(257, D, generator=gen) * 2 - 1)
        enc = RefFreq(input_dim=D, max_freq_log2=deg - 1, N_freqs=deg, log_sampling=True)
        y = enc(x)
        assert y.shape == (257, D + 2 * D * deg)
        out[f"x_{D}_{deg}"] = x.numpy(); out[f"y_{D}_{deg}"] = y.numpy()
    save("freq_encoder", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "freq":
        gen_freq()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "e2e":
        gen_e2e("b1", 1)
        gen_e2e("b2", 2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "editgrid":
        gen_editgrid()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "get_rays":
        gen_get_rays()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg0":
        gen_cfg0()
        sys.exit(0)
    gen_get_rays()
    gen_editgrid()
    gen_grid_offsets()
    gen_ffmlp_init()
    gen_mlp_chain()
    gen_run_path()
    gen_cfg0()
    gen_ops()
    gen_e2e("b1", 1)
    gen_e2e("b2", 2)
    gen_run_upsample()
    gen_density_grid()
    gen_freq()
