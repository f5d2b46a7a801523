"""The zero-edit drop-in train step: what a LAENeRF checkout runs once `backend.install_as_reference_backends()` has put
the HIP backend under the reference's four extension names -- and nothing else of this repository.

This module restates, as measurement harness, the CALLERS' side of the drop-in boundary: the allocate-then-call rules of the
reference's operator wrappers and the operator sequence of its training step, going through the backend modules by their
REFERENCE names and argument lists only (no `_ex` forms, no `blc`, no shadow table, no fused head / criterion / optimizer,
no HIP graph).  It lives under tools/ on purpose: nothing here is product code or used by the product path (`laenerf_amd/renderer.py`,
`network.py`, `optim.py`); `bench.py`'s
`drop_in_step` times it beside the fused headline and `tests/test_gpu_dropin.py` checks that both compute the same step.

    raymarching/raymarching.py:19-49,161-291    near_far_from_aabb, march_rays_train (three torch.zeros sample buffers, torch.rand
                                                noise, D2H read of the counter while mean_count <= 0), composite_rays_train
                                                (zeros_like gradients)
    gridencoder/grid.py:24-93,145-161           per-call `embeddings.to(half)` under autocast, [L,B,C] output + permute/reshape
                                                view, backward: [B,L,C] -> [L,B,C] copy + zeros_like(table) gradient
    shencoder/sphere_harmonics.py:14-86         fp32 in / out
    ffmlp/ffmlp.py:15-86,150-168                custom_fwd(cast_inputs=half), `pad = 128 - B % 128` rows of zeros ALWAYS appended,
                                                forward_buffer [layers,B,hidden] empty, backward_buffer zeros, zeros_like grads
    activation.py:5-17                          trunc_exp
    nerf/network_ff.py:51-79                    encoder -> sigma net -> trunc_exp -> SH -> zeros_like pad -> cat -> colour net ->
                                                two host reads (`torch.any(h.isnan()) or torch.any(h.isinf())`) -> sigmoid
    nerf/renderer.py:259-333                    run_cuda, training branch
    nerf/utils.py:535-620,1474-1482             MSELoss(reduction='none').mean(-1).mean(); scaler.scale(loss).backward();
                                                scaler.step(optimizer); scaler.update()  (torch.optim.Adam + GradScaler, eager)

Left out on purpose: `torch.autograd.set_detect_anomaly(True)` (nerf/utils.py:540, a debugging switch of the Trainer, which
is outside SURVEY 8's scope) and the Trainer's own `loss.item()` / tqdm / tensorboard lines.
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


def _mods():
    """the four extension modules under the names the reference's wrappers import (installs them on first use)"""
    import sys
    if "_raymarching" not in sys.modules or "_ffmlp" not in sys.modules:
        from laenerf_amd import backend
        backend.install_as_reference_backends()
    return sys.modules["_raymarching"], sys.modules["_gridencoder"], sys.modules["_shencoder"], sys.modules["_ffmlp"]


class Probe:
    """optional instrumentation handed to `ReferenceChain.train_loss` / `drop_in_train_step`: the chain calls `cut()` right
    before each of its host reads and `begin()` right after, so that a caller can bracket the stretches the device could run
    without waiting for the host.  The default does nothing."""

    def begin(self):
        pass

    def cut(self):
        pass


# ----------------------------------------------------------------------------------------------- operator wrappers
class _RefNearFar(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, aabb, min_near):
        rm = _mods()[0]
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        n = rays_o.shape[0]
        nears, fars = rays_o.new_empty(n), rays_o.new_empty(n)
        rm.near_far_from_aabb(rays_o, rays_d, aabb, n, min_near, nears, fars)
        return nears, fars


class _RefMarchTrain(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, bound, bitfield, C, H, nears, fars, step_counter, mean_count, perturb, align, force_all_rays,
                dt_gamma, max_steps):
        rm = _mods()[0]
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        n = rays_o.shape[0]
        sized = (not force_all_rays) and mean_count > 0
        M = n * max_steps
        if sized:
            M = mean_count + (align - mean_count % align if align > 0 else 0)
        kw = dict(dtype=rays_o.dtype, device=rays_o.device)
        xyzs, dirs, deltas = torch.zeros(M, 3, **kw), torch.zeros(M, 3, **kw), torch.zeros(M, 2, **kw)      # three memsets per call
        rays = torch.empty(n, 3, dtype=torch.int32, device=rays_o.device)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=rays_o.device)
        noises = torch.rand(n, **kw) if perturb else torch.zeros(n, **kw)
        rm.march_rays_train(rays_o, rays_d, bitfield.contiguous(), bound, dt_gamma, max_steps, n, C, H, M, nears, fars, xyzs, dirs, deltas,
                            rays, step_counter, noises)
        if not sized:                                     # the first 16 steps: a host read sizes the buffers
            m = int(step_counter[0].item())
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
            torch.cuda.empty_cache()
        return xyzs, dirs, deltas, rays


class _RefCompositeTrain(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh):
        rm = _mods()[0]
        sigmas, rgbs = sigmas.contiguous(), rgbs.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum, depth, image = sigmas.new_empty(N), sigmas.new_empty(N), sigmas.new_empty(N, 3)
        rm.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.meta = (M, N, T_thresh)
        return weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_ws, g_depth, g_image):
        rm = _mods()[0]
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.meta
        g_sigmas, g_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
        rm.composite_rays_train_backward(g_ws.contiguous(), g_image.contiguous(), sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                         T_thresh, g_sigmas, g_rgbs)
        return g_sigmas, g_rgbs, None, None, None


class _RefGridEncode(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype, align_corners, interp):
        ge = _mods()[1]
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L, C = offsets.shape[0] - 1, embeddings.shape[1]
        S, H = np.log2(per_level_scale), base_resolution
        if torch.is_autocast_enabled("cuda") and C % 2 == 0:
            embeddings = embeddings.to(torch.half)          # the whole table, every call (49 MB read, 24.5 MB written at cfg2)
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        ge.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.meta = (B, D, C, L, S, H, gridtype, interp, align_corners)
        return outputs.permute(1, 0, 2).reshape(B, L * C)   # a copy: [L,B,C] -> [B,L*C]

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        ge = _mods()[1]
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interp, align_corners = ctx.meta
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
        g_table = torch.zeros_like(embeddings)              # 24.5 MB memset per step
        g_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        ge.grid_encode_backward(grad, inputs, embeddings, offsets, g_table, B, D, C, L, S, H, dy_dx, g_inputs, gridtype, align_corners, interp)
        if g_inputs is not None:
            g_inputs = g_inputs.to(inputs.dtype)
        return g_inputs, g_table, None, None, None, None, None, None, None


class _RefSHEncode(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs):
        sh = _mods()[2]
        inputs = inputs.contiguous()
        B, D = inputs.shape
        out = torch.empty(B, degree * degree, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, D * degree * degree, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        sh.sh_encode_forward(inputs, out, B, D, degree, dy_dx)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.meta = (B, D, degree)
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        sh = _mods()[2]
        B, D, degree = ctx.meta
        g_inputs = torch.zeros_like(inputs)
        sh.sh_encode_backward(grad.contiguous(), inputs, B, D, degree, dy_dx, g_inputs)
        return g_inputs, None, None


class _RefFFMLP(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference, calc_grad_inputs):
        ff = _mods()[3]
        B = inputs.shape[0]
        inputs, weights = inputs.contiguous(), weights.contiguous()
        outputs = torch.empty(B, output_dim, device=inputs.device, dtype=inputs.dtype)
        if inference:
            scratch = torch.empty(B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
            ff.ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, scratch, outputs)
            return outputs
        fwd_buf = torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
        ff.ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, fwd_buf, outputs)
        ctx.save_for_backward(inputs, weights, outputs, fwd_buf)
        ctx.meta = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        ff = _mods()[3]
        B = grad.shape[0]
        grad = grad.contiguous()
        inputs, weights, outputs, fwd_buf = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.meta
        g_inputs = torch.zeros_like(inputs) if calc_grad_inputs else torch.zeros(1, device=grad.device, dtype=grad.dtype)
        g_weights = torch.zeros_like(weights)
        bwd_buf = torch.zeros(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
        ff.ffmlp_backward(grad, inputs, weights, fwd_buf, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                          calc_grad_inputs, bwd_buf, g_inputs, g_weights)
        return (g_inputs if calc_grad_inputs else None), g_weights, None, None, None, None, None, None, None, None


class _RefTruncExp(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


# ----------------------------------------------------------------------------------------------- modules
class _RefFFMLPModule(nn.Module):
    """parameters and call rule of ffmlp.py:99-168 (flat fp32 `weights`, seed-42 init, 128-row padding that is always added)"""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers):
        super().__init__()
        self.input_dim, self.output_dim, self.hidden_dim, self.num_layers = input_dim, output_dim, hidden_dim, num_layers
        self.padded_output_dim = 16 * math.ceil(output_dim / 16)
        self.weights = nn.Parameter(torch.zeros(hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)))
        torch.manual_seed(42)
        self.weights.data.uniform_(-math.sqrt(3 / hidden_dim), math.sqrt(3 / hidden_dim))
        _mods()[3].allocate_splitk(num_layers + 1)

    def forward(self, x):
        B, C = x.shape
        pad = 128 - B % 128
        x = torch.cat([x, torch.zeros(pad, C, dtype=x.dtype, device=x.device)], dim=0)
        out = _RefFFMLP.apply(x, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers, 0, 6,
                              not self.training, x.requires_grad)
        return out[:B, :self.output_dim]


class ReferenceChain(nn.Module):
    """`NeRFNetwork` of nerf/network_ff.py on `NeRFRenderer.run_cuda`'s training branch, through the reference-named backends"""

    def __init__(self, bound=1, min_near=0.2, density_scale=1.0, nan_check=True, fused_head=False):
        """fused_head (one of INTEGRATION.md 3b's optional edits, off for the zero-edit step): `laenerf_amd.ffmlp.nerf_head` in the
        place of network_ff.py:57-79 (sigma net, trunc_exp, SH, cat, colour net, the NaN check, sigmoid)"""
        super().__init__()
        self.fused_head = fused_head
        from laenerf_amd.gridencoder.grid import level_offsets
        self.bound, self.min_near, self.density_scale, self.nan_check = bound, min_near, density_scale, nan_check
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.per_level_scale = np.exp2(np.log2(2048 * bound / 16) / 15)
        offsets = level_offsets(3, 16, self.per_level_scale, 16, 19, False)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), 2).uniform_(-1e-4, 1e-4))
        self.sigma_net = _RefFFMLPModule(32, 16, 64, 2)
        self.color_net = _RefFFMLPModule(32, 3, 64, 3)
        self.register_buffer("aabb_train", torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32))
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        self.mean_count, self.local_step = 0, 0

    def get_params(self, lr):
        return [{"params": [self.embeddings], "lr": lr}, {"params": list(self.sigma_net.parameters()), "lr": lr},
                {"params": [], "lr": lr}, {"params": list(self.color_net.parameters()), "lr": lr}]

    def update_mean_count(self):
        """renderer.py:644-647 (the part of update_extra_state the march depends on)"""
        total = min(16, self.local_step)
        if total > 0:
            self.mean_count = int(self.step_counter[:total, 0].sum().item() / total)
        self.local_step = 0

    def network(self, x, d, probe):
        x01 = (x + self.bound) / (2 * self.bound)
        enc = _RefGridEncode.apply(x01.view(-1, 3), self.embeddings, self.offsets, self.per_level_scale, 16, x01.requires_grad, 0, False, 0)
        if self.fused_head:
            from laenerf_amd.ffmlp import nerf_head
            return nerf_head(enc, d, self.sigma_net.weights, self.color_net.weights)
        h = self.sigma_net(enc)
        sigma = _RefTruncExp.apply(h[..., 0])
        geo = h[..., 1:]
        sh = _RefSHEncode.apply((d / 1).reshape(-1, 3), 4, d.requires_grad)
        h = torch.cat([sh, geo, torch.zeros_like(geo[..., :1])], dim=-1)
        h = self.color_net(h)
        if self.nan_check:
            probe.cut()
            bad = bool(torch.any(h.isnan())) or bool(torch.any(h.isinf()))      # two host reads per forward (network_ff.py:72)
            probe.begin()
            if bad:
                print("nan/inf detected")
        return sigma, torch.sigmoid(h)

    def render_train(self, rays_o, rays_d, bg_color=1, perturb=True, T_thresh=1e-4, max_steps=1024, dt_gamma=0, probe=None):
        probe = probe or Probe()
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        nears, fars = _RefNearFar.apply(rays_o, rays_d, self.aabb_train, self.min_near)
        counter = self.step_counter[self.local_step % 16]
        counter.zero_()
        self.local_step += 1
        if self.mean_count <= 0:
            probe.cut()
        xyzs, dirs, deltas, rays = _RefMarchTrain.apply(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears,
                                                        fars, counter, self.mean_count, perturb, 128, False, dt_gamma, max_steps)
        if self.mean_count <= 0:
            probe.begin()
        sigmas, rgbs = self.network(xyzs, dirs, probe)
        sigmas = self.density_scale * sigmas
        weights_sum, depth, image = _RefCompositeTrain.apply(sigmas, rgbs, deltas, rays, T_thresh)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return {"image": image, "depth": depth, "weights_sum": weights_sum, "n_rows": xyzs.shape[0]}

    def train_loss(self, rays_o, rays_d, gt, probe=None):
        out = self.render_train(rays_o, rays_d, probe=probe)
        loss = torch.nn.functional.mse_loss(out["image"], gt, reduction="none").mean(-1)
        return loss.mean(), out


def one_edit_train_step(chain, fused_adam, batch, probe=None):
    """the same step after ONE of INTEGRATION.md 3b's edits: `laenerf_amd.optim.FusedAdam` in the place of torch.optim.Adam +
    GradScaler (nerf/utils.py:1474-1478 becomes `opt.backward(opt.scale(loss)); opt.step()`); everything else as in
    drop_in_train_step.  No host read is left in the optimizer (found_inf stays on the device)."""
    probe = probe or Probe()
    with torch.autocast("cuda", dtype=torch.float16):
        loss, out = chain.train_loss(*batch, probe=probe)
    fused_adam.backward(fused_adam.scale(loss))
    fused_adam.step()
    return loss, out


def drop_in_train_step(chain, optimizer, scaler, batch, probe=None, split_unscale=False):
    """one step of the reference's loop (nerf/utils.py:1472-1478).  split_unscale: `scaler.unscale_(optimizer)` as a call of its
    own before `scaler.step` (same kernels; lets a probe separate the stretch before GradScaler's host read from the one after)"""
    probe = probe or Probe()
    optimizer.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        loss, out = chain.train_loss(*batch, probe=probe)
    scaler.scale(loss).backward()
    if split_unscale:
        scaler.unscale_(optimizer)
    probe.cut()                                             # scaler.step reads found_inf on the host before optimizer.step()
    scaler.step(optimizer)
    scaler.update()
    return loss, out
