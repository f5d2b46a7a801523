"""Real spherical-harmonics direction encoding on the HIP backend -- the operator interface of the reference's
`shencoder/sphere_harmonics.py` (`sh_encode(inputs, degree, calc_grad_inputs)`, `SHEncoder(input_dim, degree)`).

[B,3] unit directions -> [B, degree^2] basis values in the reference's fixed order (shencoder.cu:50-120), fp32.
`calc_grad_inputs` additionally stores d(basis)/d(direction) for the backward (never needed for ray directions).
"""
import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

from ..backend import shencoder_backend as _backend


class SHEncodeFn(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, dirs, degree, want_input_grad=False):
        dirs = dirs.contiguous()
        n, dim = dirs.shape
        width = degree * degree
        basis = dirs.new_empty(n, width)
        jac = dirs.new_empty(n, dim * width) if want_input_grad else None
        _backend.sh_encode_forward(dirs, basis, n, dim, degree, jac)
        ctx.save_for_backward(dirs, jac)
        ctx.shape = (n, dim, degree)
        return basis

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_basis):
        dirs, jac = ctx.saved_tensors
        if jac is None:                                  # forward was asked not to keep the Jacobian
            return None, None, None
        n, dim, degree = ctx.shape
        grad_dirs = torch.zeros_like(dirs)               # the kernel accumulates (shencoder.cu:358-382)
        _backend.sh_encode_backward(grad_basis.contiguous(), dirs, n, dim, degree, jac, grad_dirs)
        return grad_dirs, None, None


def sh_encode(inputs, degree, calc_grad_inputs=False):
    return SHEncodeFn.apply(inputs, degree, calc_grad_inputs)


class SHEncoder(nn.Module):
    """sphere_harmonics.py:61-86: degree 1..8, three input dimensions; `forward(inputs, size=1)` divides by `size` first"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        if input_dim != 3:
            raise AssertionError("SH encoder only support input dim == 3")
        if not 0 < degree <= 8:
            raise AssertionError("SH encoder only supports degree in [1, 8]")
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = degree * degree

    def extra_repr(self):
        return f"input_dim={self.input_dim}, degree={self.degree}"

    def forward(self, inputs, size=1):
        scaled = inputs / size
        lead = scaled.shape[:-1]
        flat = scaled.reshape(-1, self.input_dim)
        return sh_encode(flat, self.degree, flat.requires_grad).reshape(*lead, self.output_dim)
