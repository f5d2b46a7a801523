"""Encoder + head as ONE autograd op (MI355X-native): everything `NeRFNetwork.forward` does (nerf/network_ff.py:51-81).

Between the hash-grid kernels and the MLP kernels the features stay in the grid kernels' native level-major layout
[16, M, 2]: the forward needs no [L,B,C] -> [B,L*C] transpose (grid.py:57 does it with a permute copy) and the backward
hands the MLP's input gradient to the grid backward without the inverse one (grid.py:75).  Numerically identical to
`GridEncoder` followed by `nerf_head` (same kernels, same arithmetic; only the addressing differs).
"""
import numpy as np
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from .backend import ffmlp_backend as _mlp
from .backend import gridencoder_backend as _grid
from .ffmlp.head import COLOR_NET_PARAMS, SIGMA_NET_PARAMS


class _nerf_field(Function):
    calls = 0               # forwards through the fused op since import (tests assert WHICH path rendered)

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, dirs, embeddings, sigma_weights, color_weights, enc, sigma_shadow, color_shadow, bound, density_scale, plan=None):
        _nerf_field.calls += 1
        M = x.shape[0]
        L = enc.num_levels
        x = x.float().contiguous()
        dirs = dirs.float().contiguous()
        table = enc.shadow.table_half(embeddings) if enc.shadow is not None else embeddings.to(torch.half)
        shadows = (sigma_shadow, color_shadow) if sigma_shadow is not None and color_shadow is not None else None
        if shadows is not None:
            ws, wc = sigma_shadow.table_half(sigma_weights), color_shadow.table_half(color_weights)
        else:
            ws, wc = sigma_weights.half().contiguous(), color_weights.half().contiguous()
        in_map = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))
        S, H = np.log2(enc.per_level_scale), enc.base_resolution
        feats = torch.empty(L, M, 2, device=x.device, dtype=torch.half)                 # level-major
        _grid.grid_encode_forward(x, table, enc.offsets, feats, M, 3, 2, L, S, H, None, enc.gridtype_id, enc.align_corners,
                                  enc.interp_id, blc=False, in_map=in_map, offsets_host=enc.offsets_host)
        h = torch.empty(M, 16, device=x.device, dtype=torch.half)
        sigmas = torch.empty(M, device=x.device, dtype=torch.float32)
        rgbs = torch.empty(M, 3, device=x.device, dtype=torch.float32)
        _mlp.nerf_head_forward(feats, dirs, ws, wc, M, density_scale, h, sigmas, rgbs, level_major=True)
        ctx.save_for_backward(x, dirs, table, ws, wc, feats, h, rgbs)
        ctx.enc, ctx.shadows, ctx.in_map, ctx.geom = enc, shadows, in_map, (M, L, S, H)
        ctx.density_scale = density_scale
        ctx.plan = plan
        ctx.wdtypes = (sigma_weights.dtype, color_weights.dtype)
        return sigmas, rgbs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_sigmas, grad_rgbs):
        x, dirs, table, ws, wc, feats, h, rgbs = ctx.saved_tensors
        enc = ctx.enc
        M, L, S, H = ctx.geom
        grad_sigmas = grad_sigmas.float().contiguous()
        grad_rgbs = grad_rgbs.float().contiguous()
        grad_h = torch.empty_like(h)
        grad_feats = torch.empty_like(feats)                                              # level-major, like feats
        wflag = None
        if ctx.shadows is not None:
            gws, gwc = ctx.shadows[0].grad_half, ctx.shadows[1].grad_half
            wflag = _weights_flag(ctx.shadows)
        else:
            gws, gwc = torch.empty_like(ws), torch.empty_like(wc)
        _mlp.nerf_head_backward(grad_sigmas, grad_rgbs, feats, dirs, h, rgbs, ws, wc, M, ctx.density_scale, grad_h, grad_feats,
                                gws, gwc, accumulate=ctx.shadows is not None, level_major=True, nonfinite_flag=wflag)
        grad_table = enc.shadow.grad_half if enc.shadow is not None else torch.zeros_like(table)
        flag = enc.shadow.flag_for_backward(M) if enc.shadow is not None else None    # the optimizer's found_inf word, or None
        touched = enc.shadow.touched_for_backward(M) if flag is not None else None     # its "ever touched" bitmap, or None
        if ctx.plan is not None and touched is not None and not getattr(ctx.plan, "marks_touched", False):
            touched = None                                 # a plan made without the bitmap: nothing was marked
        if enc.shadow is not None and touched is None:
            enc.shadow.mark_all_touched()
        _grid.grid_encode_backward(grad_feats, x, table, enc.offsets, grad_table, M, 3, 2, L, S, H, None, None, enc.gridtype_id,
                                   enc.align_corners, enc.interp_id, blc=False, in_map=ctx.in_map, offsets_host=enc.offsets_host, plan=ctx.plan,
                                   nonfinite_flag=flag, touched_lines=touched)
        return (None, None, None if enc.shadow is not None else grad_table,
                None if ctx.shadows is not None else gws.to(ctx.wdtypes[0]),
                None if ctx.shadows is not None else gwc.to(ctx.wdtypes[1]), None, None, None, None, None, None)


def _weights_flag(shadows):
    """one flag for the two weight accumulators the head backward writes: the optimizer's found_inf word if both shadows
    carry the same one, else None (and both writes count as unreported)"""
    a, b = shadows[0].flag_for_backward(), shadows[1].flag_for_backward()
    if a is not None and a == b:
        return a
    shadows[0].unreported = shadows[1].unreported = True
    return None


def field_supported(enc, sigma_net, color_net):
    return (enc.input_dim == 3 and enc.level_dim == 2 and enc.num_levels == 16
            and sigma_net.weights.numel() == SIGMA_NET_PARAMS and color_net.weights.numel() == COLOR_NET_PARAMS)


def nerf_field(x, dirs, enc, sigma_net, color_net, bound=1, density_scale=1.0, plan=None):
    """x [M,3] in [-bound, bound], dirs [M,3] unit -> sigmas [M] fp32, rgbs [M,3] fp32 (fp16 table and MLPs; M % 16 == 0).
    plan: `field_backward_plan(x, enc, bound)` computed earlier (positions only) -- the table-gradient pass then skips
    its counting half."""
    if x.shape[0] % 16 != 0:
        raise RuntimeError("nerf_field: the number of samples must be a multiple of 16")
    return _nerf_field.apply(x, dirs, enc.embeddings, sigma_net.weights, color_net.weights, enc, sigma_net.shadow,
                             color_net.shadow, float(bound), float(density_scale), plan)


@torch.no_grad()
def field_backward_plan(x, enc, bound=1):
    """counting half of the hash-grid backward for the samples x [M,3] (fp16 table path): depends on the positions
    only, so it can be launched right after the march, on another stream, beside the forward pass"""
    x = x.float().contiguous()
    if x.shape[0] > (1 << 24):          # beyond the binned pipeline's batch limit (include/laenerf.h): no plan, the backward
        return None                     # then runs whole on the generic path
    in_map = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))
    touched = enc.shadow.touched_for_backward(x.shape[0]) if enc.shadow is not None else None     # the optimizer's bitmap, or None
    return _grid.grid_backward_plan(x, enc.offsets, x.shape[0], 3, 2, enc.num_levels, np.log2(enc.per_level_scale), enc.base_resolution,
                                    enc.gridtype_id, enc.align_corners, enc.interp_id, True, in_map, offsets_host=enc.offsets_host,
                                    touched_lines=touched)


@torch.no_grad()
def field_density(x, enc, sigma_net, bound=1, density_scale=1.0):
    """sigma [M] fp32 of the points x [M,3] -- what the occupancy-grid maintenance asks of NeRFNetwork.density
    (nerf/renderer.py:594, 625: only `sigma` is read).  Encoder output stays level-major (no [M,32] transpose: 52 us per
    2.1 M-cell sweep), the sigma net's h is not stored (67 MB per sweep)."""
    M = x.shape[0]
    Mp = (M + 15) // 16 * 16
    x = x.float().contiguous()
    if Mp != M:
        x = torch.cat([x, x.new_zeros(Mp - M, 3)], dim=0)
    table = enc.shadow.table_half(enc.embeddings) if enc.shadow is not None else enc.embeddings.detach().to(torch.half)
    ws = sigma_net.shadow.table_half(sigma_net.weights) if sigma_net.shadow is not None else sigma_net.weights.detach().half().contiguous()
    in_map = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))
    L = enc.num_levels
    feats = torch.empty(L, Mp, 2, device=x.device, dtype=torch.half)
    _grid.grid_encode_forward(x, table, enc.offsets, feats, Mp, 3, 2, L, np.log2(enc.per_level_scale), enc.base_resolution, None,
                              enc.gridtype_id, enc.align_corners, enc.interp_id, blc=False, in_map=in_map, offsets_host=enc.offsets_host)
    sigmas = torch.empty(Mp, device=x.device, dtype=torch.float32)
    _mlp.nerf_density_forward(feats, ws, Mp, density_scale, None, sigmas, level_major=True)
    return sigmas[:M]
