"""`trunc_exp`, the density activation of the NeRF networks (reference activation.py:5-17).

Forward: exp in fp32 whatever the autocast state.  Backward: the incoming gradient times exp of the input clamped to
[-15, 15], so one exploding logit cannot turn the whole step into inf.  (The fused NeRF head applies the same rule inside
its kernels, csrc/ffmlp.hip; this autograd function serves the operator-by-operator path.)
"""
import torch
from torch.amp import custom_bwd, custom_fwd

_CLAMP = 15.0


class TruncExp(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits):
        ctx.save_for_backward(logits)
        return logits.exp()

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        logits, = ctx.saved_tensors
        return grad_out * logits.clamp(min=-_CLAMP, max=_CLAMP).exp()


def trunc_exp(x):
    return TruncExp.apply(x)
