"""Host-side mirror of the reference's `gridencoder/grid.py` (GridEncoder / grid_encode) on the HIP backend.

Public behaviour identical to the reference (names, defaults, parameter/buffer names so checkpoints load:
`embeddings [sO, C]`, `offsets [L+1] int32`, grid.py:128-134).  Internally the HIP kernels read/write the
`[B, L*C]` layout directly, which removes the permute+reshape copy of grid.py:57 and the
`.permute(1,0,2).contiguous()` of grid.py:75.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from ..backend import gridencoder_backend as _backend

_gridtype_to_id = {"hash": 0, "tiled": 1}
_interp_to_id = {"linear": 0, "smoothstep": 1}


class _grid_encode(Function):
    """grid.py:24-89"""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0, shadow=None, in_map=(0.0, 1.0), offsets_host=None):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)
        H = base_resolution
        # autocast: half-precision table, float coordinates; odd C stays float (grid.py:41-44)
        ctx.shadow = None
        if torch.is_autocast_enabled("cuda") and C % 2 == 0:
            if shadow is not None:                     # fp16 table + fp16 gradient accumulator kept by the fused optimizer
                embeddings = shadow.table_half(embeddings)
                ctx.shadow = shadow
            else:
                embeddings = embeddings.to(torch.half)
        outputs = torch.empty(B, L * C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype,
                                     align_corners, interpolation, blc=True, in_map=in_map, offsets_host=offsets_host)
        ctx.offsets_host = offsets_host
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation]
        ctx.in_map = in_map
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        grad = grad.contiguous()                       # [B, L*C], consumed in place by the _blc kernel
        if grad.dtype != embeddings.dtype:
            grad = grad.to(embeddings.dtype)
        # with a shadow the gradient is accumulated straight into the optimizer's persistent fp16 buffer (zeroed by the
        # optimizer after it has consumed it) and autograd sees no gradient for `embeddings`
        grad_embeddings = ctx.shadow.grad_half if ctx.shadow is not None else torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        flag = ctx.shadow.flag_for_backward(B) if ctx.shadow is not None and D == 3 and C == 2 else None
        touched = ctx.shadow.touched_for_backward(B) if flag is not None and grad.dtype == torch.half else None
        if ctx.shadow is not None and (flag is None or touched is None):
            ctx.shadow.unreported = ctx.shadow.unreported or flag is None
            ctx.shadow.mark_all_touched()
        _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx,
                                      grad_inputs, gridtype, ctx.align_corners, interpolation, blc=True, in_map=ctx.in_map,
                                      offsets_host=ctx.offsets_host, nonfinite_flag=flag, touched_lines=touched)
        if dy_dx is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
            if ctx.in_map[1] != 1.0:
                grad_inputs = grad_inputs * ctx.in_map[1]          # chain rule of the folded affine map
        return (grad_inputs, (None if ctx.shadow is not None else grad_embeddings), None, None, None, None, None, None, None,
                None, None, None)


class TableShadow:
    """fp16 copy of a GridEncoder table + persistent fp16 gradient accumulator, owned by `laenerf_amd.optim.FusedAdam`
    (replaces the per-step `embeddings.to(half)` of grid.py:43-44 and the `zeros_like` of grid.py:77)"""

    def __init__(self, embeddings):
        self.half = embeddings.detach().to(torch.half).contiguous()
        # the accumulator is a view of a flat store padded (with zeros that stay zero) to a multiple of 840 = lcm(1..8)
        # elements: a data-parallel reduce-scatter over any W <= 8 ranks takes the store as it is (dist.allreduce_gradients)
        n = self.half.numel()
        self.grad_store = torch.zeros((n + 839) // 840 * 840, dtype=torch.half, device=self.half.device)
        self.grad_half = self.grad_store[:n].view_as(self.half)
        self.version = embeddings._version
        # FusedAdam's found_inf word (device address) once the optimizer has adopted the table and every level goes through
        # the binned backward: that backward then reports the non-finite values it stores itself, and the optimizer leaves
        # the table out of its scan (optim.FusedAdam.step).  `unreported` = grad_half was written by something that does
        # not report (a folded .grad, a gradient all-reduce): the next step scans it again.
        self.nonfinite_flag = None
        self.unreported = False
        # "ever touched" bitmap (one bit per 8 entries = one 64-byte line of the fp32 table), set by the binned backward where it
        # stores a gradient; the optimizer skips lines whose bit is clear (gradient and both Adam moments exactly zero).  A
        # writer that does not report turns every bit on (mark_all_touched): from then on nothing is skipped.
        self.touched_lines = None

    def mark_all_touched(self):
        if self.touched_lines is not None:
            self.touched_lines.fill_(-1)

    def flag_for_backward(self, n_samples=0):
        """address to hand to grid_encode_backward(nonfinite_flag=...), or None (then the write counts as unreported)"""
        if self.nonfinite_flag is None or n_samples > (1 << 24):     # beyond the binned pipeline's batch limit
            self.unreported = True
            self.mark_all_touched()
            return None
        return self.nonfinite_flag

    def touched_for_backward(self, n_samples=0):
        """address of the bitmap for grid_encode_backward(touched_lines=...), or None"""
        if self.touched_lines is None or self.nonfinite_flag is None or n_samples > (1 << 24):
            return None
        return self.touched_lines.data_ptr()

    def table_half(self, embeddings):
        if embeddings._version != self.version:        # someone else wrote the fp32 table (load_state_dict, init, ...)
            self.half.copy_(embeddings.detach())
            self.version = embeddings._version
        return self.half


grid_encode = _grid_encode.apply


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """level sizes of grid.py:118-127: min(2^T, (res[+1])^D) rounded up to a multiple of 8"""
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


class GridEncoder(nn.Module):
    """grid.py:96-161"""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:     # overrides per_level_scale (grid.py:101-102)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        # the same numbers on the host: handed to the library with every call (balances the forward's level -> XCD
        # schedule, lets the backward skip an idle launch; include/laenerf.h `offsets_host`).  The sizes are fixed by the
        # constructor arguments (grid.py:118-127), so this copy cannot go stale.
        self.offsets_host = offsets.copy()
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.shadow = None                                 # TableShadow once a FusedAdam owns the table
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)         # grid.py:138-140

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> "
                f"{int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} "
                f"gridtype={self.gridtype} align_corners={self.align_corners} interpolation={self.interpolation}")

    def forward(self, inputs, bound=1):
        # [-bound, bound] -> [0, 1] (grid.py:149) happens inside the kernels as (x + bound) * fp32(1 / (2 * bound)), which
        # is how torch evaluates `(inputs + bound) / (2 * bound)` on the GPU: no separate add / div kernels
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        in_map = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id, self.shadow, in_map,
                              self.offsets_host)
        return outputs.view(prefix_shape + [self.output_dim])

    def attach_shadow(self):
        """called by FusedAdam: keep an fp16 table + gradient accumulator next to the fp32 parameter"""
        if self.level_dim % 2 != 0 or not self.embeddings.is_cuda:
            raise RuntimeError("GridEncoder.attach_shadow: needs an even level_dim (fp16 path, grid.py:41-44) on the GPU")
        self.shadow = TableShadow(self.embeddings)
        return self.shadow

    @torch.amp.autocast("cuda", enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """grid.py:164-184: adds the TV gradient to embeddings.grad (call between backward() and step())"""
        D = self.input_dim
        C = self.embeddings.shape[1]
        L = self.offsets.shape[0] - 1
        S = np.log2(self.per_level_scale)
        H = self.base_resolution
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        _backend.grad_total_variation(inputs.contiguous(), self.embeddings, self.embeddings.grad, self.offsets, weight, B,
                                      D, C, L, S, H, self.gridtype_id, self.align_corners)
