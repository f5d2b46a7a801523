"""NeRF positional (frequency) encoding on the HIP backend -- the operator interface of the reference's
`freqencoder/freq.py` (`freq_encode(inputs, degree, output_dim)`, `FreqEncoder(input_dim, degree)`).

Row layout of the result: the D inputs, then for every frequency f < degree the D sines followed by the D cosines of
x * 2^f (freqencoder.cu:30-58).  Always evaluated in fp32 (freq.py:17).
"""
import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

from ..backend import freqencoder_backend as _backend


def encoded_width(input_dim, degree):
    return input_dim * (1 + 2 * degree)


class FreqEncodeFn(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, degree, width):
        x = x.contiguous()
        n, dim = x.shape
        if width != encoded_width(dim, degree):
            raise RuntimeError(f"freq_encode: output_dim must be {encoded_width(dim, degree)} for input_dim={dim}, degree={degree}")
        y = x.new_empty(n, width)
        _backend.freq_encode_forward(x, n, dim, degree, width, y)
        ctx.save_for_backward(y)                         # the backward needs the sines / cosines only (freqencoder.cu:63-94)
        ctx.shape = (n, dim, degree, width)
        return y

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_y):
        y, = ctx.saved_tensors
        n, dim, degree, width = ctx.shape
        grad_x = y.new_empty(n, dim)                     # every element is written by the kernel
        _backend.freq_encode_backward(grad_y.contiguous(), y, n, dim, degree, width, grad_x)
        return grad_x, None, None


def freq_encode(inputs, degree, output_dim):
    return FreqEncodeFn.apply(inputs, degree, output_dim)


class FreqEncoder(nn.Module):
    """`get_encoder('frequency', multires=degree)` (encoding.py:59-62)"""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = encoded_width(input_dim, degree)

    def extra_repr(self):
        return f"input_dim={self.input_dim}, degree={self.degree}, output_dim={self.output_dim}"

    def forward(self, inputs, **_):
        lead = inputs.shape[:-1]
        flat = inputs.reshape(-1, self.input_dim)
        return freq_encode(flat, self.degree, self.output_dim).reshape(*lead, self.output_dim)
