"""Fused NeRF head: everything `NeRFNetwork.forward` does after the grid encoder (nerf/network_ff.py:57-79) as one
forward kernel and two backward kernels (MI355X-native; see `lae_nerf_head_forward` in include/laenerf.h).

    sigma, rgb = nerf_head(enc, dirs, sigma_net.weights, color_net.weights)

is numerically the composition  sigma_net -> trunc_exp / geo_feat -> SH(4) -> cat -> color_net -> sigmoid  of the
separate operators (same fp16 roundings at the same places; `tests/test_gpu_ffmlp.py` checks it against that
composition and against the CPU oracle), without any intermediate tensor in HBM except h [M,16] fp16.
"""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from ..backend import ffmlp_backend as _backend

SIGMA_NET_PARAMS = 64 * (32 + 64 + 16)          # FFMLP(32 -> 64 -> 64 -> 16)
COLOR_NET_PARAMS = 64 * (32 + 2 * 64 + 16)      # FFMLP(32 -> 64 -> 64 -> 64 -> 16)


class _nerf_head(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, enc, dirs, sigma_weights, color_weights, density_scale, sigma_shadow=None, color_shadow=None):
        M = enc.shape[0]
        if enc.shape[1] != 32 or sigma_weights.numel() != SIGMA_NET_PARAMS or color_weights.numel() != COLOR_NET_PARAMS:
            raise RuntimeError("nerf_head: needs the 32-wide encoder, FFMLP(32,64,2 layers,16) and FFMLP(32,64,3 layers,16)")
        if M % 16 != 0:
            raise RuntimeError("nerf_head: the number of samples must be a multiple of 16 (march_rays_train aligns to 128)")
        enc = enc.half().contiguous()
        dirs = dirs.float().contiguous()
        # with a FusedAdam attached, the fp16 weights are kept by the optimizer (no per-step cast) and the weight
        # gradients are added into its persistent fp16 buffers (autograd sees none)
        ctx.shadows = (sigma_shadow, color_shadow) if sigma_shadow is not None and color_shadow is not None else None
        if ctx.shadows is not None:
            ws, wc = sigma_shadow.table_half(sigma_weights), color_shadow.table_half(color_weights)
        else:
            ws, wc = sigma_weights.half().contiguous(), color_weights.half().contiguous()
        h = torch.empty(M, 16, device=enc.device, dtype=torch.half)
        sigmas = torch.empty(M, device=enc.device, dtype=torch.float32)
        rgbs = torch.empty(M, 3, device=enc.device, dtype=torch.float32)
        _backend.nerf_head_forward(enc, dirs, ws, wc, M, density_scale, h, sigmas, rgbs)
        ctx.save_for_backward(enc, dirs, ws, wc, h, rgbs)
        ctx.density_scale = density_scale
        ctx.need_enc_grad = ctx.needs_input_grad[0]
        ctx.wdtypes = (sigma_weights.dtype, color_weights.dtype)
        return sigmas, rgbs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_sigmas, grad_rgbs):
        enc, dirs, ws, wc, h, rgbs = ctx.saved_tensors
        M = enc.shape[0]
        grad_sigmas = grad_sigmas.float().contiguous()
        grad_rgbs = grad_rgbs.float().contiguous()
        grad_h = torch.empty_like(h)
        grad_enc = torch.empty_like(enc) if ctx.need_enc_grad else None
        if ctx.shadows is not None:
            a, b = ctx.shadows[0].flag_for_backward(), ctx.shadows[1].flag_for_backward()    # the optimizer's found_inf word
            flag = a if a is not None and a == b else None
            if flag is None:
                ctx.shadows[0].unreported = ctx.shadows[1].unreported = True
            _backend.nerf_head_backward(grad_sigmas, grad_rgbs, enc, dirs, h, rgbs, ws, wc, M, ctx.density_scale, grad_h,
                                        grad_enc, ctx.shadows[0].grad_half, ctx.shadows[1].grad_half, accumulate=True,
                                        nonfinite_flag=flag)
            return grad_enc, None, None, None, None, None, None
        gws, gwc = torch.empty_like(ws), torch.empty_like(wc)
        _backend.nerf_head_backward(grad_sigmas, grad_rgbs, enc, dirs, h, rgbs, ws, wc, M, ctx.density_scale, grad_h,
                                    grad_enc, gws, gwc)
        return grad_enc, None, gws.to(ctx.wdtypes[0]), gwc.to(ctx.wdtypes[1]), None, None, None


def nerf_head(enc, dirs, sigma_weights, color_weights, density_scale=1.0, sigma_shadow=None, color_shadow=None):
    """enc [M,32] (grid-encoder output), dirs [M,3] unit fp32 -> sigmas [M] fp32, rgbs [M,3] fp32"""
    return _nerf_head.apply(enc, dirs, sigma_weights, color_weights, float(density_scale), sigma_shadow, color_shadow)


@torch.no_grad()
def nerf_density(enc, sigma_weights, density_scale=1.0, want_geo_feat=True):
    """inference-only density query after the encoder (network_ff.py:83-96): enc [M,32] -> sigma [M] fp32 and
    (optionally) h [M,16] fp16 whose columns 1..15 are geo_feat.  M is padded to a multiple of 16 here."""
    M = enc.shape[0]
    if enc.shape[1] != 32 or sigma_weights.numel() != SIGMA_NET_PARAMS:
        raise RuntimeError("nerf_density: needs the 32-wide encoder and FFMLP(32,64,2 layers,16)")
    enc = enc.half().contiguous()
    Mp = (M + 15) // 16 * 16
    if Mp != M:
        enc = torch.cat([enc, enc.new_zeros(Mp - M, 32)], dim=0)
    sigmas = torch.empty(Mp, device=enc.device, dtype=torch.float32)
    h = torch.empty(Mp, 16, device=enc.device, dtype=torch.half) if want_geo_feat else None
    _backend.nerf_density_forward(enc, sigma_weights.half().contiguous(), Mp, density_scale, h, sigmas)
    return sigmas[:M], (h[:M] if want_geo_feat else None)
