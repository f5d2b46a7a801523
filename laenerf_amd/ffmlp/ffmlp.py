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
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False):
        B = inputs.shape[0]
        inputs, weights = inputs.contiguous(), weights.contiguous()
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
        grad_weights = torch.empty_like(weights)
        backward_buffer = None if ctx.fused else torch.empty(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
        _backend.ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers,
                                activation, output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights)
        return (grad_inputs if calc_grad_inputs else None), grad_weights, None, None, None, None, None, None, None, None


ffmlp_forward = _ffmlp_forward.apply

_ACT = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}


def convert_activation(act):
    """ffmlp.py:89-96 (anything unknown, e.g. 'none', maps to 6)"""
    return _ACT.get(act, 6)


class FFMLP(nn.Module):
    """ffmlp.py:99-168: bias-free MLP, `weights` is one flat fp32 parameter laid out
    W0[hidden,in] | W1..[hidden,hidden] | Wout[16,hidden]."""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16
        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"
        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.shadow = None                                      # fp16 weights + gradient buffer once a FusedAdam owns them
        self.reset_parameters()
        _backend.allocate_splitk(self.num_layers + 1)

    def attach_shadow(self):
        from ..gridencoder.grid import TableShadow
        if not self.weights.is_cuda:
            raise RuntimeError("FFMLP.attach_shadow: weights must be on the GPU")
        self.shadow = TableShadow(self.weights)
        return self.shadow

    def cleanup(self):
        _backend.free_splitk()

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)                                   # ffmlp.py:141-144
        std = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-std, std)

    def forward(self, inputs):
        B, C = inputs.shape
        # The reference always pads to the next multiple of 128, +128 rows when already aligned (ffmlp.py:157-159),
        # because its kernel owns 128-row tiles.  The MFMA kernel owns 16-row tiles, so an aligned batch needs no
        # padded copy; the rows returned to the caller are identical either way.
        pad = (16 - B % 16) % 16
        if pad > 0:
            inputs = torch.cat([inputs, torch.zeros(pad, C, dtype=inputs.dtype, device=inputs.device)], dim=0)
        outputs = ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim,
                                self.num_layers, self.activation, self.output_activation, not self.training,
                                inputs.requires_grad)
        if B != outputs.shape[0] or self.padded_output_dim != self.output_dim:
            outputs = outputs[:B, :self.output_dim]
        return outputs
