"""Fused Adam + GradScaler for the NeRF parameters.

The reference's step is `scaler.scale(loss).backward(); scaler.step(Adam); scaler.update()` (nerf/utils.py:1474-1482)
with `torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15)` (main_nerf.py:223).  `FusedAdam` reproduces exactly that
arithmetic (same formulas, same skip / backoff / growth rules) in three kernels of csrc/optimizer.hip, keeps all state
on the device (HIP-graph capturable) and owns, for every `GridEncoder` table, the fp16 shadow the encoder gathers from
and the fp16 gradient accumulator the encoder's backward adds into:

    opt = FusedAdam(model, lr=1e-2)                    # or FusedAdam(model, param_groups=model.get_params(lr))
    loss = ...; opt.scale(loss).backward(); opt.step()  # no zero_grad needed: gradients are consumed and zeroed

`state_dict()` / `load_state_dict()` use torch.optim.Adam's layout (`exp_avg`, `exp_avg_sq`, `step`).
"""
import ctypes

import torch

from . import _lib
from .ffmlp import FFMLP
from .gridencoder import GridEncoder


class FusedAdam:
    def __init__(self, model, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0, param_groups=None,
                 grad_scaler=True, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        groups = param_groups if param_groups is not None else [{"params": [p for p in model.parameters() if p.requires_grad], "lr": lr}]
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.use_scaler = bool(grad_scaler)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        tables = {id(m.embeddings): m for m in model.modules() if isinstance(m, GridEncoder) and m.level_dim % 2 == 0}
        tables.update({id(m.weights): m for m in model.modules() if isinstance(m, FFMLP)})
        self.param_groups = []
        self.items = []                                   # (param, exp_avg, exp_avg_sq, shadow or None, group index)
        dev = None
        for gi, g in enumerate(groups):
            params = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append({"params": params, "lr": float(g.get("lr", lr))})
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors on the GPU (no CPU fallback)")
                dev = p.device
                shadow = None
                owner = tables.get(id(p))
                if owner is not None:
                    shadow = owner.attach_shadow()         # fp16 copy + fp16 gradient accumulator next to the parameter
                else:
                    p.grad = torch.zeros_like(p)           # persistent fp32 gradient (stable address for graph replay)
                self.items.append((p, torch.zeros_like(p), torch.zeros_like(p), shadow, gi))
        # device state block (include/laenerf.h): scale | tracker | found_inf | skip | step | 1/bc1 | sqrt(bc2) | 1/scale | skipped
        self.state = torch.zeros(16, dtype=torch.int32, device=dev)
        self._scale_view = self.state.view(torch.float32)
        self._scale_view[0] = init_scale if self.use_scaler else 1.0
        self.lrs = torch.tensor([g["lr"] for g in self.param_groups], dtype=torch.float32, device=dev)
        if len(self.items) > 8:
            raise RuntimeError("FusedAdam: at most 8 parameter tensors (one multi-tensor launch)")
        self._args = None                                  # ctypes pointer tables of the multi-tensor kernels

    # ---- GradScaler face
    def scale(self, loss):
        return loss * self._scale_view[0] if self.use_scaler else loss

    def backward(self, loss):
        """`loss.backward()` with a cached root gradient: autograd's implicit `ones_like(loss)` is a fill launch of its own,
        and in a captured step that one-block kernel sits on the critical path (~19 us between the criterion and the
        compositing backward, profiles/r1m)."""
        key = (tuple(loss.shape), loss.dtype, str(loss.device))
        cache = self.__dict__.setdefault("_root_grads", {})
        one = cache.get(key)
        if one is None:
            one = cache[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
        loss.backward(gradient=one)

    def get_scale(self):
        return float(self._scale_view[0].item())

    @property
    def steps_taken(self):
        return int(self.state[4].item())

    @property
    def steps_skipped(self):
        return int(self.state[8].item())

    def set_lr(self, lr, group=None):
        """learning-rate schedule hook: writes device memory, so a captured graph picks the new value up"""
        for gi, g in enumerate(self.param_groups):
            if group is None or gi == group:
                g["lr"] = float(lr)
                self.lrs[gi] = float(lr)

    def sync_shadows(self):
        """re-derive every fp16 shadow table from its fp32 parameter.  In-place writes to a parameter (load_state_dict,
        `with torch.no_grad(): p.copy_(...)`) are noticed automatically through the tensor version counter; writes
        through `p.data` are not visible to it -- call this after them."""
        for p, _, _, shadow, _ in self.items:
            if shadow is not None:
                shadow.half.copy_(p.detach())
                shadow.version = p._version

    def zero_grad(self, set_to_none=False):
        """gradients are zeroed by step(); provided for Trainer code that calls it anyway"""
        for p, _, _, shadow, _ in self.items:
            if shadow is not None:
                shadow.grad_half.zero_()
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _grad(self, p, shadow):
        if shadow is not None:
            if p.grad is not None:
                # the parameter was also used through a path that does not know the shadow (fp32 encoder call, FFMLP
                # module forward with autograd): fold that gradient into the accumulator
                shadow.grad_half.add_(p.grad.to(torch.half))
                p.grad = None
            return shadow.grad_half, 1
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
            raise RuntimeError("FusedAdam: fp32 contiguous .grad expected")
        return p.grad, 0

    def _tables(self):
        grads = [self._grad(p, sh) for p, _, _, sh, _ in self.items]
        key = tuple(g.data_ptr() for g, _ in grads)
        if self._args is None or self._args["key"] != key:
            n = len(self.items)
            arr = lambda vals: (ctypes.c_void_p * n)(*vals)
            self._args = {
                "key": key, "n": n,
                "grads": arr(key), "is_half": (ctypes.c_int * n)(*[h for _, h in grads]),
                "sizes": (ctypes.c_uint64 * n)(*[p.numel() for p, *_ in self.items]),
                "params": arr([p.data_ptr() for p, *_ in self.items]),
                "m": arr([m.data_ptr() for _, m, *_ in self.items]), "v": arr([v.data_ptr() for _, _, v, *_ in self.items]),
                "shadows": arr([None if sh is None else sh.half.data_ptr() for _, _, _, sh, _ in self.items]),
                "lrs": arr([self.lrs.data_ptr() + 4 * gi for *_, gi in self.items]),
            }
        return self._args

    @torch.no_grad()
    def step(self):
        """check (any non-finite gradient?) -> begin (skip / step decision, scale update) -> apply: three launches"""
        lib, st, s = _lib.load(), self.state.data_ptr(), _lib.stream()
        a = self._tables()
        if self.use_scaler:
            _lib.check(lib.lae_adam_check_multi(a["n"], a["grads"], a["is_half"], a["sizes"], st, s), "adam_check")
        _lib.check(lib.lae_adam_begin(st, self.betas[0], self.betas[1], self.growth_interval, self.growth_factor,
                                      self.backoff_factor, int(self.use_scaler), s), "adam_begin")
        _lib.check(lib.lae_adam_apply_multi(a["n"], a["params"], a["m"], a["v"], a["grads"], a["is_half"], a["shadows"],
                                            a["sizes"], a["lrs"], st, self.betas[0], self.betas[1], self.eps,
                                            self.weight_decay, s), "adam_apply")

    # ---- torch.optim.Adam-compatible checkpoints
    def state_dict(self):
        step = self.steps_taken
        return {"state": {i: {"step": torch.tensor(float(step)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
                          for i, (_, m, v, _, _) in enumerate(self.items)},
                "param_groups": [{"lr": g["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}
                                 for g in self.param_groups],
                "scaler": {"scale": self.get_scale(), "growth_tracker": int(self.state[1].item())}}

    def load_state_dict(self, sd):
        for i, (_, m, v, _, _) in enumerate(self.items):
            e = sd["state"][i]
            m.copy_(e["exp_avg"]); v.copy_(e["exp_avg_sq"])
            self.state[4] = int(e["step"])
        for gi, g in enumerate(sd.get("param_groups", [])):
            self.set_lr(g["lr"], gi)
        if "scaler" in sd:
            self._scale_view[0] = sd["scaler"]["scale"]
            self.state[1] = sd["scaler"]["growth_tracker"]
