"""Host-side mirror of the reference's `raymarching/raymarching.py` on the HIP backend.

Same public names, argument order, defaults, output shapes and in-place behaviour as the
reference (cited per function), so `nerf/renderer.py` / `editing/*` written against the
reference call these unchanged.  Every output/workspace is allocated here and handed to the
backend, exactly like the reference's autograd.Functions do.  GPU tensors only: there is no
CPU fallback (the reference's wrappers likewise `.cuda()` everything, raymarching.py:34-35).
"""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from ..backend import raymarching_backend as _backend

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train", "march_rays", "march_rays_distill", "composite_rays", "composite_rays_distill",
           "compact_rays_alive", "render_frame", "composite_rays_train_blend", "composite_rays_train_blend_mse", "density_grid_positions", "density_grid_partial_positions", "density_grid_update", "mark_untrained_grid"]


def _gpu(t):
    return t if t.is_cuda else t.cuda()


def _rays(t):
    return _gpu(t).contiguous().view(-1, 3)


class _near_far_from_aabb(Function):
    """raymarching.py:19-49"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        _backend.near_far_from_aabb(rays_o, rays_d, _gpu(aabb).contiguous(), N, min_near, nears, fars)
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    """raymarching.py:52-80"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, radius):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        _backend.sph_from_ray(rays_o, rays_d, radius, N, coords)
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    """raymarching.py:83-104: coords [N,3] int in [0,128) -> indices [N] int32"""

    @staticmethod
    def forward(ctx, coords):
        coords = _gpu(coords)
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        _backend.morton3D(coords.int().contiguous(), N, indices)
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    """raymarching.py:106-126"""

    @staticmethod
    def forward(ctx, indices):
        indices = _gpu(indices)
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        _backend.morton3D_invert(indices.int().contiguous(), N, coords)
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    """raymarching.py:129-155: grid [C, H^3] float -> bitfield [C*H^3/8] uint8"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, grid, thresh, bitfield=None):
        grid = _gpu(grid).contiguous()
        N = grid.shape[0] * grid.shape[1] // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        _backend.packbits(grid, N, thresh, bitfield)
        return bitfield


packbits = _packbits.apply


def _round_up_always(m, align):
    """the reference's `m += align - m % align` (adds a full `align` when already aligned)"""
    return m + (align - m % align) if align > 0 else m


class _march_rays_train(Function):
    """raymarching.py:161-235.  Differences, all invisible to callers: rows of `rays` come out in ray-id
    order with offsets = exclusive scan of counts (the reference's order depends on atomic arrival)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        density_bitfield = _gpu(density_bitfield).contiguous()
        dev, dt = rays_o.device, rays_o.dtype
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            M = _round_up_always(mean_count, align)
        # rows no ray owns are zero-filled by the kernel (the reference's torch.zeros, raymarching.py:207-209)
        xyzs = torch.empty(M, 3, dtype=dt, device=dev)
        dirs = torch.empty(M, 3, dtype=dt, device=dev)
        deltas = torch.empty(M, 2, dtype=dt, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        rows_end = torch.empty(1, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, dtype=dt, device=dev) if perturb else torch.zeros(N, dtype=dt, device=dev)
        _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M,
                                  nears.contiguous(), fars.contiguous(), xyzs, dirs, deltas, rays, step_counter, noises, rows_end)
        rays.rows_end = rows_end                         # consumed by composite_rays_train_blend (not part of the reference API)
        if force_all_rays or mean_count <= 0:
            m = _round_up_always(int(step_counter[0].item()), align)      # D2H sync, first 16 steps only
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    """raymarching.py:238-291"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        _backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is not propagated (raymarching.py:275)
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _backend.composite_rays_train_backward(grad_weights_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs,
                                               deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


class _composite_rays_train_blend(Function):
    """MI355X-native: composite_rays_train + the post-ops of run_cuda (nerf/renderer.py:321, 325) in the same kernels:
    image + (1 - weights_sum) * bg_color and clamp(depth - nears, min=0) / (fars - nears).  The backward writes every
    gradient row itself (no zero fills).  `rays` must come from this package's march_rays_train (ray-id order)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays, nears, fars, bg_rays, bg, rows_end, T_thresh):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev, dt = sigmas.device, sigmas.dtype
        weights_sum, depth, image = (torch.empty(N, dtype=dt, device=dev), torch.empty(N, dtype=dt, device=dev),
                                     torch.empty(N, 3, dtype=dt, device=dev))
        depth_out, image_out = torch.empty(N, dtype=dt, device=dev), torch.empty(N, 3, dtype=dt, device=dev)
        _backend.composite_rays_train_forward_blend(sigmas, rgbs, deltas, rays, M, N, T_thresh, nears.contiguous(),
                                                    fars.contiguous(), bg_rays, bg, weights_sum, depth, image, depth_out, image_out)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image, bg_rays, rows_end)
        ctx.dims = [M, N, T_thresh, bg]
        ctx.mark_non_differentiable(depth_out)
        ctx.set_materialize_grads(False)                 # an unused weights_sum arrives as None, not as a zero fill
        return weights_sum, depth_out, image_out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, deltas, rays, weights_sum, image, bg_rays, rows_end = ctx.saved_tensors
        M, N, T_thresh, bg = ctx.dims
        if grad_image is None and grad_weights_sum is None:
            return None, None, None, None, None, None, None, None, None, None
        if grad_weights_sum is None:
            grad_weights_sum = torch.zeros_like(weights_sum)
        if grad_image is None:
            grad_image = torch.zeros_like(image)
        grad_sigmas, grad_rgbs = torch.empty_like(sigmas), torch.empty_like(rgbs)
        _backend.composite_rays_train_backward_blend(grad_weights_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs,
                                                     deltas, rays, weights_sum, image, M, N, T_thresh, bg_rays, bg, rows_end,
                                                     grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None, None, None, None, None, None


def _bg_args(bg_color, device):
    bg_rays, bg = None, (0.0, 0.0, 0.0)
    if torch.is_tensor(bg_color) and bg_color.numel() > 3:
        bg_rays = bg_color.to(device, torch.float32).reshape(-1, 3).contiguous()
    elif torch.is_tensor(bg_color):
        v = [float(x) for x in bg_color.reshape(-1).tolist()]
        bg = tuple(v * 3 if len(v) == 1 else v)
    elif isinstance(bg_color, (int, float)):
        bg = (float(bg_color),) * 3
    else:
        bg = tuple(float(x) for x in bg_color)
    return bg_rays, bg


def composite_rays_train_blend(sigmas, rgbs, deltas, rays, nears, fars, bg_color=1, T_thresh=1e-4):
    """-> weights_sum [N], depth normalised to [0,1] [N], image blended over bg_color [N,3]
    bg_color: number, 3 numbers / tensor of 3, or a per-ray [N,3] tensor (renderer.py:313-321)"""
    rows_end = getattr(rays, "rows_end", None)
    if rows_end is None:
        raise RuntimeError("composite_rays_train_blend: `rays` must be the tensor returned by laenerf_amd march_rays_train")
    bg_rays, bg = _bg_args(bg_color, sigmas.device)
    return _composite_rays_train_blend.apply(sigmas, rgbs, deltas, rays, nears, fars, bg_rays, bg, rows_end, T_thresh)


# root gradients known to be all ones (laenerf_amd.optim.FusedAdam.backward registers the tensor it passes to
# loss.backward): for them the fused node below hands its stored sample gradients on unchanged.  address -> weak reference:
# an address whose tensor has died may belong to anything by now
_unit_root_grads = {}


def register_unit_root_grad(t):
    import weakref
    _unit_root_grads[t.data_ptr()] = weakref.ref(t)


def _is_unit_root_grad(t):
    r = _unit_root_grads.get(t.data_ptr())
    return r is not None and r() is not None and r().shape == t.shape


class _composite_rays_train_blend_mse(Function):
    """composite_rays_train_blend + the trainer's criterion and loss scaling (`MSELoss(pred_rgb, gt).mean()` then
    `scaler.scale(loss)`, nerf/utils.py train_step) as ONE autograd node and ONE kernel: d loss / d pixel of a ray depends on
    that ray's pixel only, so the compositing forward, the criterion and the compositing backward of a ray run back to back
    in the wavefront that owns it (`lae_composite_rays_train_step`; three launches and two kernel boundaries in the middle of
    the step before).  The sample gradients are therefore computed in forward() for an upstream gradient of 1 and stored;
    backward() returns them -- multiplied by the upstream gradient unless that is known to be ones."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, deltas, rays, nears, fars, bg_rays, bg, rows_end, T_thresh, target, scale, defer_loss=False):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev, dt = sigmas.device, sigmas.dtype
        weights_sum, depth, image = (torch.empty(N, dtype=dt, device=dev), torch.empty(N, dtype=dt, device=dev),
                                     torch.empty(N, 3, dtype=dt, device=dev))
        depth_out, image_out = torch.empty(N, dtype=dt, device=dev), torch.empty(N, 3, dtype=dt, device=dev)
        target = target.float().contiguous()
        if target.shape != image_out.shape:
            raise RuntimeError("composite_rays_train_blend_mse: target must be [N,3]")
        out = torch.empty(2, dtype=torch.float32, device=dev)
        grad_image = torch.empty_like(image_out)
        grad_sigmas, grad_rgbs = torch.empty_like(sigmas), torch.empty_like(rgbs)
        partials = torch.empty((N + 3) // 4, dtype=torch.float32, device=dev)
        _backend.composite_rays_train_step(sigmas, rgbs, deltas, rays, M, N, T_thresh, nears.contiguous(), fars.contiguous(), bg_rays,
                                           bg, rows_end, target, scale, weights_sum, depth, image, depth_out, image_out, grad_image,
                                           grad_sigmas, grad_rgbs, out, partials, defer_loss=defer_loss)
        ctx.save_for_backward(grad_sigmas, grad_rgbs)
        ctx.mark_non_differentiable(weights_sum, depth_out, image_out, out)
        ctx.set_materialize_grads(False)                 # no zero-filled gradients for the four auxiliary outputs (4 fill launches)
        return out[0], weights_sum, depth_out, image_out, out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_loss, *_):
        if grad_loss is None:
            return (None,) * 13
        grad_sigmas, grad_rgbs = ctx.saved_tensors
        if not _is_unit_root_grad(grad_loss):              # a general upstream gradient: d(loss) scales every sample gradient
            gl = grad_loss.float()
            scaled = grad_sigmas * gl
            from ..backend import retarget_pending_loss
            retarget_pending_loss(grad_sigmas, scaled)     # a deferred loss value follows the tensor the head backward will receive
            grad_sigmas, grad_rgbs = scaled, grad_rgbs * gl
        return grad_sigmas, grad_rgbs, None, None, None, None, None, None, None, None, None, None, None


def composite_rays_train_blend_mse(sigmas, rgbs, deltas, rays, nears, fars, target, bg_color=1, T_thresh=1e-4, scaler=None,
                                   defer_loss=None):
    """-> (loss, weights_sum, depth, image): loss = MSE(image, target) times the loss scale of `scaler` (a FusedAdam, a
    1-element fp32 cuda tensor, or None); `loss.unscaled` holds the plain MSE.  Only `loss` carries a gradient.
    defer_loss (default: True when `scaler` is a FusedAdam): the VALUE of loss / loss.unscaled is NaN until the backward pass
    has run (the fused head's backward sums it in its reduction launch; FusedAdam.backward() / step() finish it otherwise) --
    the gradients do not depend on it, and the trainer reads it after the step (nerf/utils.py `loss.item()` for logging)."""
    rows_end = getattr(rays, "rows_end", None)
    if rows_end is None:
        raise RuntimeError("composite_rays_train_blend_mse: `rays` must be the tensor returned by laenerf_amd march_rays_train")
    bg_rays, bg = _bg_args(bg_color, sigmas.device)
    scale = None
    if scaler is not None:
        scale = scaler if torch.is_tensor(scaler) else (scaler._scale_view[:1] if scaler.use_scaler else None)
    if defer_loss is None:
        defer_loss = scaler is not None and not torch.is_tensor(scaler) and hasattr(scaler, "finish_loss")
    loss, weights_sum, depth, image, both = _composite_rays_train_blend_mse.apply(sigmas, rgbs, deltas, rays, nears, fars, bg_rays, bg,
                                                                                  rows_end, T_thresh, target, scale, bool(defer_loss))
    loss.unscaled = both[1]
    return loss, weights_sum, depth, image


def _infer_buffers(n_alive, n_step, align, dt, dev):
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    return (M, torch.zeros(M, 3, dtype=dt, device=dev), torch.zeros(M, 3, dtype=dt, device=dev),
            torch.zeros(M, 2, dtype=dt, device=dev))


class _march_rays(Function):
    """raymarching.py:297-348"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        dev, dt = rays_o.device, rays_o.dtype
        M, xyzs, dirs, deltas = _infer_buffers(n_alive, n_step, align, dt, dev)
        noises = torch.rand(n_alive, dtype=dt, device=dev) if perturb else torch.zeros(n_alive, dtype=dt, device=dev)
        _backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                            density_bitfield, near, far, xyzs, dirs, deltas, noises)
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _march_rays_distill(Function):
    """raymarching.py:355-411"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, edit_bitfield, C, H,
                near, far, align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        dev, dt = rays_o.device, rays_o.dtype
        M, xyzs, dirs, deltas = _infer_buffers(n_alive, n_step, align, dt, dev)
        edit_occ = torch.zeros(M, dtype=torch.bool, device=dev)
        noises = torch.rand(n_alive, dtype=dt, device=dev) if perturb else torch.zeros(n_alive, dtype=dt, device=dev)
        _backend.march_rays_distill(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C,
                                    H, density_bitfield, edit_bitfield, near, far, xyzs, dirs, deltas, edit_occ, noises)
        return xyzs, dirs, deltas, edit_occ


march_rays_distill = _march_rays_distill.apply


class _composite_rays(Function):
    """raymarching.py:413-435 (in place on rays_alive, rays_t, weights_sum, depth, image)"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        _backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(),
                                deltas, weights_sum, depth, image)
        return tuple()


composite_rays = _composite_rays.apply


class _composite_rays_distill(Function):
    """raymarching.py:437-461"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, weights_edit_sum, depth,
                depth_edit, image, int_edit, T_thresh=1e-2):
        _backend.composite_rays_distill(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.contiguous(),
                                        rgbs.contiguous(), deltas, weights_sum, weights_edit_sum, depth, depth_edit,
                                        int_edit, image)
        return tuple()


composite_rays_distill = _composite_rays_distill.apply


def compact_rays_alive(rays_alive, n_alive=None):
    """MI355X-native replacement for `rays_alive = rays_alive[rays_alive >= 0]` (renderer.py:375): stable
    device-side compaction.  Returns (out_alive [n_alive] int32, n_out device int32[1]); the caller may keep
    n_out on the device (no sync) or read it."""
    n_alive = rays_alive.shape[0] if n_alive is None else n_alive
    out = torch.empty(max(n_alive, 1), dtype=torch.int32, device=rays_alive.device)
    n_out = torch.empty(1, dtype=torch.int32, device=rays_alive.device)
    _backend.compact_rays_alive(rays_alive, n_alive, out, n_out)
    return out, n_out


@torch.no_grad()
def render_frame(rays_o, rays_d, aabb, min_near, density_bitfield, bound, C, H, table_half, offsets, per_level_scale,
                 base_resolution, sigma_weights_half, color_weights_half, edit_bitfield=None, gridtype_id=0, align_corners=False,
                 interp_id=0, density_scale=1.0, dt_gamma=0, max_steps=1024, T_thresh=1e-4, max_n_step=8, row_budget=0,
                 noises=None, bg_color=None, scale_depth=True, want_stats=False, offsets_host=None):
    """MI355X-native: the inference loop of NeRFRenderer.run_cuda (nerf/renderer.py:335-387; run_cuda_distill :394-480
    when `edit_bitfield` is given) as ONE backend call -- loop state on the device, no host sync per iteration.
    Same per-ray arithmetic and iteration schedule as march_rays / network / composite_rays called in the Python loop.

    row_budget: rows (samples) one iteration may put through the network; 0 = N, the reference's rule
    `n_step = max(min(N // n_alive, 8), 1)`.  A larger budget means fewer, larger iterations (same per-ray samples).
    bg_color: None (no blend, as run_cuda_distill), a scalar / 3 numbers, or a [N,3] tensor.
    Returns dict(image [N,3], depth [N], weights_sum [N]) (+ weights_edit, depth_edit; + stats when want_stats)."""
    import numpy as np
    rays_o, rays_d = _rays(rays_o), _rays(rays_d)
    N, dev = rays_o.shape[0], rays_o.device
    weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
    depth = torch.empty(N, dtype=torch.float32, device=dev)
    image = torch.empty(N, 3, dtype=torch.float32, device=dev)
    weights_edit = depth_edit = None
    if edit_bitfield is not None:
        weights_edit = torch.empty(N, dtype=torch.float32, device=dev)
        depth_edit = torch.empty(N, dtype=torch.float32, device=dev)
    bg_rays, bg_rgb = None, (0.0, 0.0, 0.0)
    if torch.is_tensor(bg_color) and bg_color.numel() == 3 * N and N > 1:
        bg_rays = _gpu(bg_color).float().reshape(N, 3).contiguous()
    elif bg_color is not None:
        v = [float(x) for x in (bg_color.flatten().tolist() if torch.is_tensor(bg_color) else np.atleast_1d(bg_color))]
        bg_rgb = tuple(v * 3) if len(v) == 1 else tuple(v)
    stats = _backend.render_frame(rays_o, rays_d, N, _gpu(aabb).float().contiguous(), min_near, density_bitfield, edit_bitfield,
                                  bound, dt_gamma, max_steps, C, H, table_half, offsets, offsets.shape[0] - 1,
                                  np.log2(per_level_scale), base_resolution, gridtype_id, align_corners, interp_id,
                                  sigma_weights_half, color_weights_half, density_scale, T_thresh, max_n_step, row_budget,
                                  None if noises is None else _gpu(noises).float().contiguous(), bg_rays, bg_rgb,
                                  bg_color is not None, scale_depth, weights_sum, depth, image, weights_edit, depth_edit,
                                  want_stats, offsets_host=offsets_host)
    out = {"image": image, "depth": depth, "weights_sum": weights_sum}
    if edit_bitfield is not None:
        out["weights_edit"], out["depth_edit"] = weights_edit, depth_edit
    if want_stats:
        out["stats"] = stats
    return out


# ---------------------------------------------------------------- occupancy-grid maintenance (MI355X-native kernels)
@torch.no_grad()
def density_grid_positions(n, H, bound_c, noise=None, coords=None):
    """positions + Morton indices of `update_extra_state`'s density queries (nerf/renderer.py:580-592, 602-621).
    coords None: the n = H^3 cells in meshgrid order (full sweep); else coords [n,3] int32.  noise [n,3] in [0,1)."""
    dev = (coords if coords is not None else noise).device if (coords is not None or noise is not None) else torch.device("cuda")
    xyzs = torch.empty(n, 3, dtype=torch.float32, device=dev)
    indices = torch.empty(n, dtype=torch.int32, device=dev)
    _backend.density_grid_positions(None if coords is None else _gpu(coords).int().contiguous(), n, H, bound_c,
                                    None if noise is None else _gpu(noise).float().contiguous(), xyzs, indices)
    return xyzs, indices


def density_grid_partial_positions(grid_c, coords_rand, u, H, bound_c, noise=None, rnd=None, n=None):
    """MI355X-native: the 2n query points of update_extra_state's PARTIAL sweep (nerf/renderer.py:600-621) without the host read
    of `nonzero`: coords_rand [n,3] random cells, u [n] uniform in [0,1) choosing among the occupied cells of grid_c [H^3] (> 0,
    in index order), noise [2n,3] jitter -- or rnd [2, n+1] uniforms from which both halves are drawn sorted on the device (H a
    power of two).  Returns xyzs [2n,3], indices [2n] int32 (Morton; -1 where no cell is occupied)."""
    n = coords_rand.shape[0] if rnd is None else (rnd.numel() // 2 - 1 if n is None else n)
    xyzs = torch.empty(2 * n, 3, dtype=torch.float32, device=grid_c.device)
    indices = torch.empty(2 * n, dtype=torch.int32, device=grid_c.device)
    _backend.density_grid_partial_positions(grid_c.contiguous(), None if rnd is not None else _gpu(coords_rand).int().contiguous(),
                                            None if rnd is not None else _gpu(u).float().contiguous(),
                                            None if rnd is None else _gpu(rnd).float().contiguous(), n, H, bound_c,
                                            None if noise is None else _gpu(noise).float().contiguous(), xyzs, indices)
    return xyzs, indices


@torch.no_grad()
def density_grid_update(grid_c, sigmas, indices, tmp, density_scale=1.0, decay=0.95):
    """in place on one cascade `grid_c` [H^3]: tmp[indices] = sigmas * density_scale; grid = max(grid * decay, tmp)
    on sampled, trainable cells (renderer.py:596, 627, 633-634).  tmp [H^3] int32 scratch, zero before and after."""
    _backend.density_grid_update(sigmas.float().contiguous(), indices.contiguous(), indices.numel(), density_scale, decay,
                                 grid_c.numel(), grid_c, tmp)
    return grid_c


@torch.no_grad()
def mark_untrained_grid(density_grid, poses, intrinsics, bound, min_near=0.2, filter_close_point=False, H=128):
    """in place: density_grid [C, H^3] = -1 where no training camera sees the cell (renderer.py:482-554)"""
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    poses = _gpu(torch.as_tensor(poses)).float().contiguous()
    _backend.mark_untrained_grid(poses, poses.shape[0], fx, fy, cx, cy, density_grid.shape[0], H, bound, min_near,
                                 filter_close_point, density_grid)
    return density_grid
