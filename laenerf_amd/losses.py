"""Fused training criterion: MSE + loss scaling in one kernel.

    loss = mse_loss_scaled(pred_rgb, gt_rgb, opt)      # opt: FusedAdam (its loss scale) or None
    loss.backward(); opt.step()

`loss` is the SCALED loss (what `GradScaler.scale(loss)` would return); `loss.unscaled` holds the plain MSE for
logging.  Equivalent to the reference's `criterion(pred, gt).mean(-1).mean()` followed by `scaler.scale(...)`
(nerf/utils.py train_step / train_one_epoch).
"""
import torch
from torch.autograd import Function

from . import _lib


class _mse_scaled(Function):
    @staticmethod
    def forward(ctx, pred, target, scale):
        pred = pred.float().contiguous()
        target = target.float().contiguous()
        _lib.need_cuda(pred, target, scale)
        if pred.shape != target.shape:
            raise RuntimeError("mse_loss_scaled: shapes differ")
        out = torch.empty(2, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred)
        _lib.check(_lib.load().lae_mse_loss_forward(pred.data_ptr(), target.data_ptr(), pred.numel(), _lib.ptr(scale),
                                                    out.data_ptr(), grad.data_ptr(), _lib.stream()), "mse_loss_forward")
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[0], out

    @staticmethod
    def backward(ctx, grad_out, _):
        if grad_out is None:
            return None, None, None
        (grad,) = ctx.saved_tensors
        return grad * grad_out, None, None


def mse_loss_scaled(pred, target, scaler=None):
    """scaler: a FusedAdam (uses its device-side loss scale), a 1-element fp32 cuda tensor, or None"""
    scale = None
    if scaler is not None:
        scale = scaler if torch.is_tensor(scaler) else (scaler._scale_view[:1] if scaler.use_scaler else None)
    loss, both = _mse_scaled.apply(pred, target, scale)
    loss.unscaled = both[1]
    return loss
