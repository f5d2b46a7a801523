"""Host-side mirror of the reference's `ffmlp/ffmlp.py` (FFMLP / ffmlp_forward) on the MFMA backend."""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from ..backend import ffmlp_backend as _backend


class _ffmlp_forward(Function):
    """ffmlp.py:15-83"""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False, shadow=None):
        """shadow (MI355X-native, optional): the module's TableShadow once a FusedAdam owns the weights -- the forward then reads
        the optimizer's fp16 copy (no cast launch per call) and the fused backward ADDS the weight gradient to the optimizer's
        fp16 accumulator and reports non-finite values itself (no fp16 -> fp32 `.grad`, no fold, no scan: six launches per net
        and step in the LAENeRF palette step)"""
        B = inputs.shape[0]
        ctx.shadow = None
        if shadow is not None and not inference and torch.is_autocast_enabled("cuda") and \
                _backend.fused_backward_available(input_dim, hidden_dim, num_layers, activation):
            ctx.shadow = shadow
            weights = shadow.table_half(weights)
        inputs, weights = inputs.to(torch.half).contiguous(), weights.to(torch.half).contiguous()     # (custom_fwd(cast_inputs=half) before round 5)
        outputs = torch.empty(B, output_dim, device=inputs.device, dtype=inputs.dtype)
        if not inference:
            # the fused MI355X backward recomputes the hidden activations from `inputs` (64 B/row) instead of
            # reading a [num_layers, B, hidden] buffer back from HBM, so no forward_buffer is written or kept
            fused = _backend.fused_backward_available(input_dim, hidden_dim, num_layers, activation)
            forward_buffer = None if fused else torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
            _backend.ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                   output_activation, forward_buffer, outputs)
            ctx.fused = fused
            ctx.save_for_backward(*((inputs, weights, outputs) if fused else (inputs, weights, outputs, forward_buffer)))
            ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
        else:
            _backend.ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                     output_activation, None, outputs)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        B = grad.shape[0]
        grad = grad.contiguous()
        if ctx.fused:
            (inputs, weights, outputs), forward_buffer = ctx.saved_tensors, None
        else:
            inputs, weights, outputs, forward_buffer = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.dims
        grad_inputs = torch.empty_like(inputs) if calc_grad_inputs else torch.zeros(1, device=grad.device, dtype=grad.dtype)
        backward_buffer = None if ctx.fused else torch.empty(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
        if ctx.shadow is not None:                     # straight into the optimizer's accumulator
            flag = ctx.shadow.flag_for_backward()
            _backend.ffmlp_backward(grad.to(torch.half), inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers,
                                    activation, output_activation, calc_grad_inputs, backward_buffer, grad_inputs,
                                    ctx.shadow.grad_half.view(-1), accumulate=True, nonfinite_flag=flag)
            return (grad_inputs if calc_grad_inputs else None), None, None, None, None, None, None, None, None, None, None
        grad_weights = torch.empty_like(weights)
        _backend.ffmlp_backward(grad.to(torch.half), inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers,
                                activation, output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights)
        return (grad_inputs if calc_grad_inputs else None), grad_weights, None, None, None, None, None, None, None, None, None


ffmlp_forward = _ffmlp_forward.apply

_ACT = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}


def convert_activation(act):
    """ffmlp.py:89-96 (anything unknown, e.g. 'none', maps to 6)"""
    return _ACT.get(act, 6)


class FFMLP(nn.Module):
    """The reference's fully fused MLP module (ffmlp.py:99-168) on the MFMA kernels.

    `weights`: ONE flat fp32 parameter, layers back to back, each stored [out, in] row-major:
    W0 [hidden, input_dim] | (num_layers - 1) x W [hidden, hidden] | W_out [16, hidden] (output rows beyond output_dim are
    padding).  No biases.  Same constructor constraints as the reference (AssertionError when violated)."""

    HIDDEN_CHOICES = (16, 32, 64, 128, 256)

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        assert hidden_dim in self.HIDDEN_CHOICES, f"FFMLP: hidden_dim {hidden_dim} not in {self.HIDDEN_CHOICES}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP: input_dim {input_dim} is not a positive multiple of 16"
        assert output_dim <= 16, f"FFMLP: output_dim {output_dim} exceeds the 16-wide output tile"
        assert num_layers >= 2, f"FFMLP: num_layers {num_layers} < 2 (at least three matrix products)"
        self.input_dim, self.output_dim = input_dim, output_dim
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16
        self.padded_output_dim = 16 * math.ceil(output_dim / 16)
        self.num_parameters = hidden_dim * (input_dim + (num_layers - 1) * hidden_dim + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.shadow = None                                      # fp16 weights + gradient buffer once a FusedAdam owns them
        self.reset_parameters()
        _backend.allocate_splitk(num_layers + 1)                # accepted for interface parity (ffmlp.py:126)

    def extra_repr(self):
        return (f"input_dim={self.input_dim}, output_dim={self.output_dim}, hidden_dim={self.hidden_dim}, "
                f"num_layers={self.num_layers}, activation={self.activation}")

    def reset_parameters(self):
        """U(-sqrt(3/hidden), +sqrt(3/hidden)) from generator seed 42, like the reference (ffmlp.py:141-144)"""
        torch.manual_seed(42)
        bound = math.sqrt(3.0 / self.hidden_dim)
        with torch.no_grad():
            self.weights.uniform_(-bound, bound)

    def attach_shadow(self):
        from ..gridencoder.grid import TableShadow
        if not self.weights.is_cuda:
            raise RuntimeError("FFMLP.attach_shadow: weights must be on the GPU")
        self.shadow = TableShadow(self.weights)
        return self.shadow

    def cleanup(self):
        _backend.free_splitk()

    def forward(self, inputs):
        """[B, input_dim] -> [B, output_dim].  The reference pads B to the next multiple of 128 (a further 128 rows when
        it already is one, ffmlp.py:157-159) because its kernel owns 128-row tiles; the MFMA kernel owns 16-row tiles,
        so only a ragged tail is padded.  The rows handed back are the same."""
        rows = inputs.shape[0]
        tail = -rows % 16
        if tail:
            inputs = torch.cat([inputs, inputs.new_zeros(tail, inputs.shape[1])], dim=0)
        out = ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                            self.activation, self.output_activation, not self.training, inputs.requires_grad, self.shadow)
        if tail or self.padded_output_dim != self.output_dim:
            out = out[:rows, :self.output_dim]
        return out
