"""Fused Adam + GradScaler (+ EMA) for the NeRF parameters.

The reference's step is `scaler.scale(loss).backward(); scaler.step(Adam); scaler.update()` (nerf/utils.py:1474-1482)
with `torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15)` (main_nerf.py:223), a per-step `LambdaLR` (main_nerf.py:239-245)
and a `torch_ema.ExponentialMovingAverage` of the parameters (nerf/utils.py:407-408).  `FusedAdam` reproduces exactly that
arithmetic (same formulas, same skip / backoff / growth rules) in three kernels of csrc/optimizer.hip, keeps all state
on the device (HIP-graph capturable) and owns, for every `GridEncoder` table, the fp16 shadow the encoder gathers from
and the fp16 gradient accumulator the encoder's backward adds into:

    opt = FusedAdam(model, lr=1e-2)                    # or FusedAdam(model, param_groups=model.get_params(lr))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / iters, 1))
    loss = ...; opt.scale(loss).backward(); opt.step(); sched.step()    # no zero_grad needed: gradients are consumed and zeroed

It IS a `torch.optim.Optimizer`: schedulers mutate `param_groups[i]["lr"]` and `step()` pushes a changed value to the
device; `state_dict()` / `load_state_dict()` are torch.optim.Adam's (`state[i] = {step, exp_avg, exp_avg_sq}`,
`param_groups[i]["params"]`), so a checkpoint of the reference's optimizer loads (plus a `scaler` entry).
"""
import ctypes
import os

import torch

from . import _lib
from .ffmlp import FFMLP
from .gridencoder import GridEncoder


# > 0 while a FusedAdam.backward() is running its autograd pass (see there); read by laenerf_amd.editing.style_encoder
_direct_grad = {"depth": 0}


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0, param_groups=None,
                 grad_scaler=True, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        groups = param_groups if param_groups is not None else [{"params": [p for p in model.parameters() if p.requires_grad], "lr": lr}]
        # empty groups stay (the reference's get_params has one for the parameter-free direction encoder, network_ff.py:147):
        # group indices must line up with a torch.optim.Adam checkpoint
        groups = [{**g, "params": [p for p in g["params"] if p.requires_grad]} for g in groups]
        super().__init__(groups, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.use_scaler = bool(grad_scaler)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        tables = {id(m.embeddings): m for m in model.modules() if isinstance(m, GridEncoder) and m.level_dim % 2 == 0}
        tables.update({id(m.weights): m for m in model.modules() if isinstance(m, FFMLP)})
        self.items = []                                   # (param, exp_avg, exp_avg_sq, shadow or None, group index)
        dev = None
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors on the GPU (no CPU fallback)")
                dev = p.device
                shadow = None
                owner = tables.get(id(p))
                if owner is not None:
                    shadow = owner.attach_shadow()         # fp16 copy + fp16 gradient accumulator next to the parameter
                else:
                    p.grad = torch.zeros_like(p)           # persistent fp32 gradient (stable address for graph replay)
                    p._lae_persistent_grad = True          # fused criterion nodes may add into it themselves (style_encoder._palette_point_loss)
                m, v = torch.zeros_like(p), torch.zeros_like(p)
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": m, "exp_avg_sq": v}      # torch.optim.Adam's layout
                self.items.append((p, m, v, shadow, gi))
        if dev is None:
            raise ValueError("FusedAdam: no parameter to optimise")
        # device state block (include/laenerf.h): scale | tracker | found_inf | skip | step | 1/bc1 | sqrt(bc2) | 1/scale | skipped
        self.dev_state = torch.zeros(16, dtype=torch.int32, device=dev)
        # hash-grid tables whose every level goes through the binned backward: that backward ORs found_inf (state word 2)
        # itself whenever it stores a non-finite gradient, so step() leaves them out of its scan (24.5 MB per step for the
        # 12 M-entry table) unless something else wrote their accumulator since (TableShadow.unreported)
        if self.use_scaler:
            for p, _, _, shadow, _ in self.items:
                owner = tables.get(id(p))
                if shadow is not None and isinstance(owner, GridEncoder) and owner.input_dim == 3 and owner.level_dim == 2 \
                        and owner.num_levels <= 32 and int((owner.offsets_host[1:] - owner.offsets_host[:-1]).max()) <= (1 << 21):
                    shadow.nonfinite_flag = self.dev_state.data_ptr() + 8
                    if weight_decay == 0 and not os.environ.get("LAE_ADAM_NO_TOUCHED_LINES"):   # touched-lines-only update (exact only without weight decay; the variable is an A/B switch)
                        n_words = int(_lib.load().lae_grid_touched_lines_words(p.shape[0]))
                        shadow.touched_lines = torch.zeros(n_words, dtype=torch.int32, device=p.device)
                elif shadow is not None and isinstance(owner, FFMLP):
                    shadow.nonfinite_flag = self.dev_state.data_ptr() + 8    # the fused head backward reports its weight gradients
        self._scale_view = self.dev_state.view(torch.float32)
        self._scale_view[0] = init_scale if self.use_scaler else 1.0
        self._lr_host = [float(g["lr"]) for g in self.param_groups]
        self.lrs = torch.tensor(self._lr_host, dtype=torch.float32, device=dev)
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
            from .raymarching.raymarching import register_unit_root_grad
            register_unit_root_grad(one)                   # the fused criterion node skips the multiplication by it
        # fused criterion nodes may add a small fp32 gradient straight into a parameter's persistent .grad (style_encoder.
        # _palette_point_loss) -- only inside THIS backward: torch.autograd.grad(), hooks and retain_graph users go through a plain
        # loss.backward() / autograd.grad() and get the gradient from autograd as usual
        _direct_grad["depth"] += 1
        try:
            loss.backward(gradient=one)
        finally:
            _direct_grad["depth"] -= 1
        self.finish_loss()

    @staticmethod
    def finish_loss():
        """a loss value deferred by the fused criterion (raymarching.composite_rays_train_blend_mse) that no backward kernel took
        along is summed now; no-op otherwise"""
        from .backend import flush_pending_loss
        flush_pending_loss()

    def get_scale(self):
        return float(self._scale_view[0].item())

    @property
    def steps_taken(self):
        return int(self.dev_state[4].item())

    @property
    def steps_skipped(self):
        return int(self.dev_state[8].item())

    def set_lr(self, lr, group=None):
        """learning-rate schedule hook: writes device memory, so a captured graph picks the new value up"""
        for gi, g in enumerate(self.param_groups):
            if group is None or gi == group:
                g["lr"] = float(lr)
        self.sync_lr()

    def sync_lr(self):
        """push `param_groups[i]["lr"]` (what torch's schedulers mutate) to the device copy the kernels read.  step() does
        it by itself when a value changed; a caller that REPLAYS a captured step calls this between replays (the graph
        reads the device copy, so the new rate takes effect without re-capturing)."""
        cur = [float(g["lr"]) for g in self.param_groups]
        if cur != self._lr_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: a learning rate changed inside a stream capture; call sync_lr() outside the capture")
            # staged through pinned host memory, asynchronously on the current stream: the reference's per-iteration LambdaLR
            # (main_nerf.py:239-245) changes the rate EVERY step, and a pageable host-to-device copy would block the host each
            # time.  Two alternating staging buffers + an event: a buffer is rewritten only after the copy that
            # read it has finished.
            st = self.__dict__.setdefault("_lr_stage", {"bufs": [torch.empty(len(cur), dtype=torch.float32).pin_memory() for _ in range(2)],
                                                        "evs": [None, None], "k": 0})
            k = st["k"]
            if st["evs"][k] is not None:
                st["evs"][k].synchronize()
            st["bufs"][k].copy_(torch.tensor(cur, dtype=torch.float32))
            self.lrs.copy_(st["bufs"][k], non_blocking=True)
            ev = st["evs"][k] or torch.cuda.Event()
            ev.record()
            st["evs"][k] = ev
            st["k"] = 1 - k
            self._lr_host = cur

    def sync_shadows(self):
        """re-derive every fp16 shadow table from its fp32 parameter.  In-place writes to a parameter (load_state_dict,
        `with torch.no_grad(): p.copy_(...)`) are noticed automatically through the tensor version counter; writes
        through `p.data` are not visible to it -- call this after them."""
        for p, _, _, shadow, _ in self.items:
            if shadow is not None:
                shadow.half.copy_(p.detach())
                shadow.version = p._version

    def zero_grad(self, set_to_none=False):
        """gradients are zeroed by step(); provided for Trainer code that calls it anyway.  Also clears the found_inf word:
        the binned grid backward and the fused head backward OR it at BACKWARD time, so a backward that overflowed and was
        then discarded (zero_grad without a step: dropped batch, accumulation restart) must not make the next, clean step
        be skipped.  Inside a stream capture the clear becomes part of the graph."""
        self.dev_state[2:3].zero_()
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
                if hasattr(shadow, "unreported"):
                    shadow.unreported = True             # this sum may overflow and nobody reports it: scan the table this step
                    shadow.mark_all_touched()
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
                "touched": arr([None if sh is None or getattr(sh, "touched_lines", None) is None else sh.touched_lines.data_ptr()
                                for _, _, _, sh, _ in self.items]),
            }
        return self._args

    @torch.no_grad()
    def step(self, closure=None):
        """check (any non-finite gradient?) -> begin (skip / step decision, scale update) -> apply: three launches"""
        if closure is not None:
            raise RuntimeError("FusedAdam.step: closures are not supported")
        self.sync_lr()
        self.finish_loss()
        lib, st, s = _lib.load(), self.dev_state.data_ptr(), _lib.stream()
        a = self._tables()
        for *_, shadow, _ in self.items:                     # an accumulator written by something that does not report where:
            if shadow is not None and getattr(shadow, "unreported", False) and hasattr(shadow, "mark_all_touched"):
                shadow.mark_all_touched()                    # every line counts as touched from now on
        if self.use_scaler:
            c = self._check_tables(a)
            if c["n"]:
                _lib.check(lib.lae_adam_check_multi(c["n"], c["grads"], c["is_half"], c["sizes"], st, s), "adam_check")
        _lib.check(lib.lae_adam_begin(st, self.betas[0], self.betas[1], self.growth_interval, self.growth_factor,
                                      self.backoff_factor, int(self.use_scaler), s), "adam_begin")
        _lib.check(lib.lae_adam_apply_multi(a["n"], a["params"], a["m"], a["v"], a["grads"], a["is_half"], a["shadows"],
                                            a["sizes"], a["lrs"], a["touched"], st, self.betas[0], self.betas[1], self.eps,
                                            self.weight_decay, s), "adam_apply")
        for *_, shadow, _ in self.items:                     # the accumulators are zero again
            if shadow is not None and hasattr(shadow, "unreported"):
                shadow.unreported = False

    def _check_tables(self, a):
        """the gradients step() scans for non-finite values: all of them, minus the tables whose backward reports into
        found_inf itself (see __init__)"""
        keep = tuple(not (sh is not None and getattr(sh, "nonfinite_flag", None) is not None and not sh.unreported)
                     for _, _, _, sh, _ in self.items)
        cached = a.get("check")
        if cached is None or cached["keep"] != keep:
            idx = [i for i, k in enumerate(keep) if k]
            n = len(idx)
            arr = (ctypes.c_void_p * max(n, 1))
            cached = a["check"] = {"keep": keep, "n": n, "grads": arr(*[a["grads"][i] for i in idx]),
                                   "is_half": (ctypes.c_int * max(n, 1))(*[a["is_half"][i] for i in idx]),
                                   "sizes": (ctypes.c_uint64 * max(n, 1))(*[a["sizes"][i] for i in idx])}
        return cached

    # ---- torch.optim.Adam-compatible checkpoints
    def state_dict(self):
        """torch.optim.Optimizer.state_dict() (Adam layout: state[i] = {step, exp_avg, exp_avg_sq}, param_groups with their
        `params` indices) + the GradScaler's state under "scaler" """
        step = float(self.steps_taken)
        for p, *_ in self.items:
            self.state[p]["step"] = torch.tensor(step)
        sd = super().state_dict()
        sd["scaler"] = {"scale": self.get_scale(), "growth_tracker": int(self.dev_state[1].item())}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        scaler = sd.pop("scaler", None)
        keep = {id(p): (m, v) for p, m, v, _, _ in self.items}
        super().load_state_dict(sd)
        step = 0
        for p, m, v, _, _ in self.items:                     # keep the tensors the kernels' pointer tables refer to
            e = self.state[p]
            step = int(float(e["step"]))
            m.copy_(e["exp_avg"].to(m.device)); v.copy_(e["exp_avg_sq"].to(v.device))
            e["exp_avg"], e["exp_avg_sq"] = keep[id(p)]
        self.dev_state[4] = step
        for *_, shadow, _ in self.items:                      # loaded moments may be non-zero anywhere: nothing is skippable
            if shadow is not None and hasattr(shadow, "mark_all_touched"):
                shadow.mark_all_touched()
        self._lr_host = None                                  # group learning rates came with the checkpoint
        self.sync_lr()
        if scaler is not None:
            self._scale_view[0] = scaler["scale"]
            self.dev_state[1] = scaler["growth_tracker"]


class EMA:
    """`torch_ema.ExponentialMovingAverage(parameters, decay)` of the reference's trainer (nerf/utils.py:407-408; `update()`
    once per epoch :1502-1503; `store()` / `copy_to()` / `restore()` around evaluation :1205-1215, :1252-1266) with the
    update of every tensor as ONE launch (csrc/optimizer.hip k_ema_multi).  Same arithmetic: decay_t = min(decay,
    (1 + n) / (10 + n)), shadow -= (1 - decay_t) * (shadow - param)."""

    def __init__(self, parameters, decay=0.95, use_num_updates=True):
        if not 0.0 <= decay <= 1.0:
            raise ValueError("decay must be between 0 and 1")
        self.params = [p for p in parameters if p.requires_grad]
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("EMA: parameters must be contiguous fp32 tensors on the GPU (no CPU fallback)")
        if len(self.params) > 8:
            raise RuntimeError("EMA: at most 8 parameter tensors (one multi-tensor launch)")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in self.params]
        self.collected_params = None
        n = len(self.params)
        self._n = n
        self._shadows = (ctypes.c_void_p * n)(*[t.data_ptr() for t in self.shadow_params])
        self._params = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.params])
        self._sizes = (ctypes.c_uint64 * n)(*[p.numel() for p in self.params])

    @torch.no_grad()
    def update(self):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        if [p.data_ptr() for p in self.params] != list(self._params):
            self._params = (ctypes.c_void_p * self._n)(*[p.data_ptr() for p in self.params])
        _lib.check(_lib.load().lae_ema_update_multi(self._n, self._shadows, self._params, self._sizes, float(1.0 - decay), _lib.stream()),
                   "ema_update")

    @torch.no_grad()
    def store(self):
        self.collected_params = [p.detach().clone() for p in self.params]

    @torch.no_grad()
    def copy_to(self):
        for s_, p in zip(self.shadow_params, self.params):
            p.copy_(s_)                                       # in place on the parameter: bumps its version, fp16 shadow tables follow

    @torch.no_grad()
    def restore(self):
        if self.collected_params is None:
            raise RuntimeError("EMA.restore() without a store()")
        for c, p in zip(self.collected_params, self.params):
            p.copy_(c)
        self.collected_params = None

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": [t.clone() for t in self.shadow_params],
                "collected_params": None if self.collected_params is None else [t.clone() for t in self.collected_params]}

    def load_state_dict(self, sd):
        self.decay, self.num_updates = sd["decay"], sd["num_updates"]
        for t, src in zip(self.shadow_params, sd["shadow_params"]):
            t.copy_(src.to(t.device))
        cp = sd.get("collected_params")
        self.collected_params = None if cp is None else [c.to(p.device).clone() for c, p in zip(cp, self.params)]
