"""LAENeRF's palette network on the HIP operators (reference: editing/style_encoder.py:20-256).

x [P,3] --hash grid (L=16, T=2^19)--> 32 features --weight net (64x64, ReLU)--> P_b logits --softmax--> w_hat
                                      features | SH degree 3 of d (9) | 0-pad to 48 --offset net--> 3 --tanh--> o_hat
pred = clamp(w_hat @ palette + o_hat, 0, 1)

The reference builds both MLPs with tinycudann's FullyFusedMLP (`tcnn.Network`, style_encoder.py:65-88), an
un-vendored third-party extension with no pinned version: parity is anchored on the fp32 `nn.Linear` chain of the same
shapes instead (tests/test_gpu_style.py).  Here they are `FFMLP`s (the repository's MFMA MLP; fused recompute backward
for the 32- and 48-wide inputs), and softmax / tanh / palette product / clamp and their backward are one kernel each
(csrc/palette.hip).  tcnn pads the 41-wide offset input to 48 the same way.

Not here (out of scope): the VGG style network (`StyleNetwork`, torchvision), image-space TV / depth losses.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from ..backend import style_backend as _backend
from ..encoding import get_encoder
from ..ffmlp import FFMLP


class _palette_recompose(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, w_logits, o_raw, palette, active_mask):
        M = w_logits.shape[0]
        w_logits, o_raw = w_logits.half().contiguous(), o_raw.half().contiguous()
        palette = palette.float().contiguous()
        P = palette.shape[0]
        n_active = bin(active_mask & ((1 << P) - 1)).count("1")
        dev = w_logits.device
        pred = torch.empty(M, 3, dtype=torch.half, device=dev)
        w_hat = torch.empty(M, n_active, dtype=torch.float32, device=dev)
        o_hat = torch.empty(M, 3, dtype=torch.half, device=dev)
        _backend.palette_forward(w_logits, o_raw, palette, P, active_mask, M, pred, w_hat, o_hat)
        ctx.save_for_backward(w_logits, o_raw, palette)
        ctx.meta = (P, active_mask, M)
        ctx.set_materialize_grads(False)                 # unused outputs arrive as None in backward (the kernel takes NULL)
        return pred, w_hat, o_hat

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_pred, g_w, g_o):
        w_logits, o_raw, palette = ctx.saved_tensors
        P, active_mask, M = ctx.meta
        g_pred = None if g_pred is None else g_pred.half().contiguous()
        g_w = None if g_w is None else g_w.float().contiguous()
        g_o = None if g_o is None else g_o.half().contiguous()
        g_wl, g_ol = torch.empty_like(w_logits), torch.empty_like(o_raw)
        g_pal = torch.empty_like(palette)
        _backend.palette_backward(w_logits, o_raw, palette, P, active_mask, M, g_pred, g_w, g_o, g_wl, g_ol, g_pal)
        return g_wl, g_ol, g_pal, None


class _palette_point_loss(Function):
    """recomposition + the point-wise losses of train_LAENeRF_step (nerf/utils.py:990-996) as ONE autograd node:
    forward = recomposition kernel + two reduction kernels, backward = one kernel that derives the per-point gradients of
    MSE / weight / offset terms on the fly (csrc/palette.hip k_palette_bwd<LOSS>).  The torch formulation costs ~40
    reduce / elementwise launches per step around [P, <= 16] tensors."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, w_logits, o_raw, palette, active_mask, target, lw, scale, reg_w=None):
        M = w_logits.shape[0]
        w_logits, o_raw = w_logits.half().contiguous(), o_raw.half().contiguous()
        palette_in = palette
        palette, target = palette.float().contiguous(), target.float().contiguous()
        P = palette.shape[0]
        n_active = bin(active_mask & ((1 << P) - 1)).count("1")
        dev = w_logits.device
        pred = torch.empty(M, 3, dtype=torch.half, device=dev)
        w_hat = torch.empty(M, n_active, dtype=torch.float32, device=dev)
        o_hat = torch.empty(M, 3, dtype=torch.half, device=dev)
        _backend.palette_forward(w_logits, o_raw, palette, P, active_mask, M, pred, w_hat, o_hat)
        fin = torch.empty(12, dtype=torch.float32, device=dev)
        # reg_w = (palette_loss_valid, palette_loss_distinct): the palette-only term `palet_loss` rides in the same two launches
        _backend.style_loss_forward(pred, target, w_hat, o_hat, M, n_active, lw, scale, fin,
                                    reg_palette=palette if reg_w is not None else None, reg_w=reg_w or (0.0, 0.0))
        ctx.save_for_backward(w_logits, o_raw, palette, target, fin)
        ctx.meta = (P, active_mask, M, lw, reg_w)
        # a palette parameter whose fp32 .grad is a FusedAdam's persistent buffer takes its gradient by an add inside the
        # reduction launch (round 5) instead of through autograd's AccumulateGrad (one more launch on the step's critical path)
        ctx.palette_param = palette_in if getattr(palette_in, "_lae_persistent_grad", False) and palette_in is palette else None
        ctx.mark_non_differentiable(pred, w_hat, o_hat, fin)
        ctx.set_materialize_grads(False)                 # no zero-filled gradients for the auxiliary outputs
        return fin[0], pred, w_hat, o_hat, fin

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_loss, *_):
        if g_loss is None:
            return (None,) * 8
        w_logits, o_raw, palette, target, fin = ctx.saved_tensors
        P, active_mask, M, lw, reg_w = ctx.meta
        g_wl, g_ol = torch.empty_like(w_logits), torch.empty_like(o_raw)
        owner = ctx.palette_param
        # the in-kernel add into palette.grad happens only inside FusedAdam.backward() and for a parameter nobody hooked: everyone else
        # (torch.autograd.grad, tensor / post-accumulate hooks, gradient all-reduce hooks, a second pass with retain_graph) gets the
        # gradient from autograd as usual
        from ..optim import _direct_grad
        direct = owner is not None and _direct_grad["depth"] > 0 and owner.grad is not None and owner.grad.dtype == torch.float32 \
            and owner.grad.is_contiguous() and owner.grad.shape == palette.shape and not owner._backward_hooks \
            and not getattr(owner, "_post_accumulate_grad_hooks", None)
        g_pal = owner.grad if direct else torch.empty_like(palette)
        _backend.style_loss_backward(w_logits, o_raw, palette, P, active_mask, M, target, fin, g_loss.float().reshape(1).contiguous(), lw,
                                     g_wl, g_ol, g_pal, reg_w=reg_w, accumulate=direct)
        return g_wl, g_ol, (None if direct else g_pal), None, None, None, None, None


def palette_recompose(w_logits, o_raw, palette, active_mask):
    """w_logits, o_raw [M,16] (the padded FFMLP outputs), palette [P,3], active_mask int (bit k = base k active)
    -> pred [M,3] fp16, w_hat [M, n_active] fp32, o_hat [M,3] fp16   (style_encoder.py:148-158)"""
    return _palette_recompose.apply(w_logits, o_raw, palette, int(active_mask))


class _style_features(Function):
    """hash-grid encoder + the input assembly of LAENeRF.forward_train (style_encoder.py:135-146) as ONE autograd node (round 5):
    the grid kernels' level-major features go straight into `lae_style_assemble_forward`, which writes the weight net's [Mp,32]
    rows and the offset net's [Mp,48] rows = [features | SH(d) | 0] (before: transpose launch, SH launch, `d / size`, cast, zero
    fill, cat); the backward adds the two nets' input gradients and transposes them back in one launch and hands them to the
    binned grid backward (before: two slice copies, an add, a transpose launch).  Same values bit for bit
    (tests/test_gpu_style.py::test_fused_input_assembly_equals_the_operator_chain)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, d, embeddings, enc, bound, degree, off_cols, plan=None):
        import numpy as np
        from ..backend import gridencoder_backend as _grid
        M = x.shape[0]
        Mp = (M + 15) // 16 * 16
        L = enc.num_levels
        x = x.float().contiguous()
        table = enc.shadow.table_half(embeddings) if enc.shadow is not None else embeddings.to(torch.half)
        in_map = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))
        S, H = np.log2(enc.per_level_scale), enc.base_resolution
        feats = torch.empty(L, M, 2, device=x.device, dtype=torch.half)                 # level-major
        _grid.grid_encode_forward(x, table, enc.offsets, feats, M, 3, 2, L, S, H, None, enc.gridtype_id, enc.align_corners,
                                  enc.interp_id, blc=False, in_map=in_map, offsets_host=enc.offsets_host)
        feat = torch.empty(Mp, 32, device=x.device, dtype=torch.half)
        off_in = torch.empty(Mp, off_cols, device=x.device, dtype=torch.half) if degree else None
        _backend.style_assemble_forward(feats, d.float().contiguous() if degree else None, M, Mp, degree, feat, off_in, off_cols)
        ctx.save_for_backward(x, table)
        ctx.enc, ctx.in_map, ctx.geom, ctx.off_cols, ctx.plan = enc, in_map, (M, L, S, H), off_cols, plan
        return (feat, off_in) if degree else (feat, feat)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_feat, g_off):
        from ..backend import gridencoder_backend as _grid
        x, table = ctx.saved_tensors
        enc = ctx.enc
        M, L, S, H = ctx.geom
        if g_feat is None and g_off is None:
            return (None,) * 8
        g_feat = None if g_feat is None else g_feat.to(torch.half).contiguous()
        g_off = None if g_off is None else g_off.to(torch.half).contiguous()
        off_cols = ctx.off_cols if g_off is not None and g_off.shape[1] != 32 else 32
        grad_feats = torch.empty(L, M, 2, device=x.device, dtype=torch.half)
        _backend.style_assemble_backward(g_feat, g_off, M, off_cols, grad_feats)
        grad_table = enc.shadow.grad_half if enc.shadow is not None else torch.zeros_like(table)
        flag = enc.shadow.flag_for_backward(M) if enc.shadow is not None else None
        touched = enc.shadow.touched_for_backward(M) if flag is not None else None
        if ctx.plan is not None and touched is not None and not getattr(ctx.plan, "marks_touched", False):
            touched = None                                 # a plan made without the bitmap: nothing was marked
        if enc.shadow is not None and touched is None:
            enc.shadow.unreported = enc.shadow.unreported or flag is None
            enc.shadow.mark_all_touched()
        _grid.grid_encode_backward(grad_feats, x, table, enc.offsets, grad_table, M, 3, 2, L, S, H, None, None, enc.gridtype_id,
                                   enc.align_corners, enc.interp_id, blc=False, in_map=ctx.in_map, offsets_host=enc.offsets_host,
                                   plan=ctx.plan, nonfinite_flag=flag, touched_lines=touched)
        return None, None, (None if enc.shadow is not None else grad_table), None, None, None, None, None


class LAENeRF(nn.Module):
    """style_encoder.py:20-90.  `params` needs `.bound` and `.num_palette_bases` (and `style_weight`, which must be 0:
    the VGG style network is out of scope)."""

    def __init__(self, params, encoding="hashgrid", dir_encoding=None, num_layers=3, hidden_dim=64, color_palette=None, size=256,
                 style_img=None):
        super().__init__()
        self.opt = params
        self.bound = params.bound
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * self.bound, num_levels=16, log2_hashmap_size=19)
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.num_color_bases = params.num_palette_bases
        if not 0 < self.num_color_bases <= 16:
            raise ValueError("LAENeRF: 1..16 palette bases (the MLP output tile is 16 wide)")
        if getattr(params, "style_weight", 0) > 0:
            raise NotImplementedError("LAENeRF: the VGG style network is outside the MI355X hot path (SURVEY.md 2.1)")
        self.register_buffer("active_palets", torch.ones(self.num_color_bases, dtype=torch.bool))
        self._active_mask = (1 << self.num_color_bases) - 1                    # host copy of active_palets (no sync per step)
        pal = color_palette if color_palette is not None else torch.rand(self.num_color_bases, 3, dtype=torch.float32)
        self.color_palette = nn.Parameter(pal.detach().clone().float())         # style_encoder.py:46-50 (a leaf with requires_grad)
        self.original_color_palette = None
        self.size = size
        self.dir_encoding, self.in_dim_dir = None, 0
        if dir_encoding is not None:
            self.dir_encoding, self.in_dim_dir = get_encoder(dir_encoding, degree=3)
        # tcnn FullyFusedMLP(n_hidden_layers = num_layers - 1) = in -> 64 -> 64 -> out: FFMLP with num_layers - 1 hidden GEMMs
        self.offset_in_dim = (self.in_dim + self.in_dim_dir + 15) // 16 * 16     # 41 -> 48, zero columns (tcnn pads alike)
        self.offset_net = FFMLP(self.offset_in_dim, 3, hidden_dim, num_layers - 1)
        self.weight_net = FFMLP(self.in_dim, self.num_color_bases, hidden_dim, num_layers - 1)

    # ---- the two heads as the fused MLP writes them: [M,16] fp16 with padded columns
    def plan_backward(self, x):
        """counting half of the hash-grid backward for the points x [M,3] (positions only: it can run ahead of the step, on
        another stream -- the points of a view are known before its step starts); hand the result to forward_train(x, d, plan=...) /
        forward_train_loss(..., plan=...).  None when the one-node input path does not apply (the step then plans for itself)."""
        from ..field import field_backward_plan
        if not (self.fused_inputs and x.is_cuda and x.dim() == 2 and self._encoder_fits()):
            return None
        return field_backward_plan(x, self.encoder, self.bound)

    def _encoder_fits(self):
        from ..gridencoder import GridEncoder
        enc = self.encoder
        return isinstance(enc, GridEncoder) and enc.num_levels == 16 and enc.level_dim == 2 and enc.input_dim == 3

    def _logits(self, x, d, plan=None):
        M = x.shape[0]
        fused = self._fused_inputs_ok(x, d)
        if plan is not None and not fused:
            raise RuntimeError("LAENeRF: a backward plan needs the one-node input path (half-precision hash grid under autocast)")
        if fused:
            deg = self.dir_encoding.degree if self.dir_encoding is not None else 0
            feat, off_in = _style_features.apply(x, d if deg else None, self.encoder.embeddings, self.encoder, self.bound, deg, self.offset_in_dim,
                                                 plan)
            wn, on = self.weight_net, self.offset_net
            from ..ffmlp.ffmlp import ffmlp_forward
            w_logits = ffmlp_forward(feat, wn.weights, wn.input_dim, 16, wn.hidden_dim, wn.num_layers, wn.activation, wn.output_activation,
                                     not self.training, True, wn.shadow if self.ffmlp_shadows else None)
            o_raw = ffmlp_forward(off_in, on.weights, on.input_dim, 16, on.hidden_dim, on.num_layers, on.activation, on.output_activation,
                                  not self.training, True, on.shadow if self.ffmlp_shadows else None)
            return w_logits, o_raw, M
        pad = (16 - M % 16) % 16
        feat = self.encoder(x, bound=self.bound)
        if pad:
            feat = torch.cat([feat, feat.new_zeros(pad, feat.shape[1])], 0)
        wn, on = self.weight_net, self.offset_net
        from ..ffmlp.ffmlp import ffmlp_forward
        w_logits = ffmlp_forward(feat, wn.weights, wn.input_dim, 16, wn.hidden_dim, wn.num_layers, wn.activation, wn.output_activation,
                                 not self.training, feat.requires_grad, wn.shadow if self.ffmlp_shadows else None)
        cols = [feat]
        if self.dir_encoding is not None:
            enc_d = self.dir_encoding(d).to(feat.dtype)
            if pad:
                enc_d = torch.cat([enc_d, enc_d.new_zeros(pad, enc_d.shape[1])], 0)
            cols.append(enc_d)
        width = sum(c.shape[1] for c in cols)
        if width < self.offset_in_dim:
            cols.append(feat.new_zeros(feat.shape[0], self.offset_in_dim - width))
        off_in = torch.cat(cols, -1) if len(cols) > 1 else feat
        o_raw = ffmlp_forward(off_in, on.weights, on.input_dim, 16, on.hidden_dim, on.num_layers, on.activation, on.output_activation,
                              not self.training, off_in.requires_grad, on.shadow if self.ffmlp_shadows else None)
        return w_logits, o_raw, M

    fused_inputs = True       # False: the operator chain (GridEncoder -> SHEncoder -> cast / pad / cat), e.g. for A/B tests
    ffmlp_shadows = True      # False: the MLPs' weight gradients go through autograd's fp32 .grad (the pre-round-5 path; A/B tests)

    def _fused_inputs_ok(self, x, d):
        """the one-node input assembly serves the shipped configuration: half-precision hash grid with 16 levels of 2 features under
        autocast, SH directions of degree <= 4 (or none), coordinates that need no gradient"""
        from ..gridencoder import GridEncoder
        from ..shencoder import SHEncoder
        enc = self.encoder
        if not (self.fused_inputs and x.is_cuda and torch.is_autocast_enabled("cuda") and self._encoder_fits()):
            return False
        if x.requires_grad or x.dim() != 2:
            return False
        if self.dir_encoding is None:
            return self.offset_in_dim == 32
        return isinstance(self.dir_encoding, SHEncoder) and self.dir_encoding.degree <= 4 and d is not None and not d.requires_grad \
            and 32 + self.dir_encoding.degree ** 2 <= self.offset_in_dim <= 48

    def forward_train(self, x, d=None, plan=None):
        """style_encoder.py:135-158 -> (pred_colors [M,3], w_hat [M,n_active], o_hat [M,3]); plan: `plan_backward(x)` (MI355X extension)"""
        if self.dir_encoding is not None:
            assert d is not None
        w_logits, o_raw, M = self._logits(x, d, plan)
        pred, w_hat, o_hat = palette_recompose(w_logits, o_raw, self.color_palette, self._active_mask)
        return pred[:M], w_hat[:M], o_hat[:M]

    def forward(self, x, d=None):
        """style_encoder.py:111-133"""
        return self.forward_train(x, d)[0]

    def forward_train_loss(self, x, d, target, params, scaler=None, with_palet_loss=False, plan=None):
        """MI355X-native: forward_train + the point-wise losses of train_LAENeRF_step (nerf/utils.py:990-996) in one node:
        loss = MSE(pred, target) + weights_loss(w_hat) + offset_loss(o_hat) [+ palet_loss(params) with with_palet_loss=True: the
        palette-only term and its gradient then ride in the criterion's own launches instead of ~40 tiny torch kernels per step],
        multiplied by `scaler`'s loss scale (a FusedAdam, a 1-element fp32 cuda tensor, or None).
        -> (loss, pred [M,3], w_hat, o_hat); loss.terms = [scaled loss, loss, mse, uniform, non-uniform, offset, jmax, scale, palet, ...]"""
        if self.dir_encoding is not None:
            assert d is not None
        w_logits, o_raw, M = self._logits(x, d, plan)
        if w_logits.shape[0] != M:
            raise RuntimeError("forward_train_loss: the number of points must be a multiple of 16")
        scale = None
        if scaler is not None:
            scale = scaler if torch.is_tensor(scaler) else (scaler._scale_view[:1] if scaler.use_scaler else None)
        lw = (float(params.weight_loss_uniform), float(params.weight_loss_non_uniform), float(params.offset_loss))
        reg_w = (float(params.palette_loss_valid), float(params.palette_loss_distinct)) if with_palet_loss else None
        loss, pred, w_hat, o_hat, fin = _palette_point_loss.apply(w_logits, o_raw, self.color_palette, self._active_mask, target, lw, scale, reg_w)
        loss.terms = fin
        return loss, pred, w_hat, o_hat

    def get_weights(self, x):
        """style_encoder.py:93-96"""
        feat = self.encoder(x, bound=self.bound)
        w_hat = self.weight_net(feat)[:, self.active_palets]
        return torch.softmax(w_hat, -1)

    def get_offsets(self, x, d):
        """style_encoder.py:98-109 (raw offsets, no tanh)"""
        return self._logits(x, d)[1][:x.shape[0], :3]

    @torch.no_grad()
    def distill_color_palettes(self, x_terms, n=10, thresh=0.025):
        """style_encoder.py:160-173: bases whose mean weight over n sampled views is below `thresh` are switched off.
        x_terms: list of [P_i,3] point sets (the reference indexes its EditDataset)."""
        idx = torch.randint(0, len(x_terms), (n,))
        weights = torch.zeros(self.num_color_bases, dtype=torch.float32, device=self.color_palette.device)
        for i in idx.tolist():
            weights[self.active_palets] += self.get_weights(x_terms[i].to(weights.device)).float().mean(0)
        self.set_active_palets(weights / n >= thresh)

    def set_active_palets(self, mask):
        mask = torch.as_tensor(mask, dtype=torch.bool, device=self.active_palets.device)
        if not bool(mask.any()):
            raise ValueError("LAENeRF: at least one palette base must stay active")
        self.active_palets.copy_(mask)
        self._active_mask = sum(1 << k for k, on in enumerate(mask.tolist()) if on)

    def get_color_palette(self):
        return self.color_palette[self.active_palets]

    def set_color_palette(self, palet):
        if self.original_color_palette is None:
            self.original_color_palette = self.color_palette.detach().clone()
        with torch.no_grad():
            self.color_palette[self.active_palets] = palet

    # ---- point-wise losses of train_LAENeRF_step (nerf/utils.py:993-996; style_encoder.py:183-205)
    def weights_loss(self, pred_bary_weights, params):
        uniform_loss = torch.sum(pred_bary_weights, dim=0).max()
        non_uniform_loss = (1 - pred_bary_weights.max(dim=-1).values).sum()
        return uniform_loss * params.weight_loss_uniform + non_uniform_loss * params.weight_loss_non_uniform

    def palet_loss(self, params):
        dists = (torch.pow(self.color_palette[:, None, :] - self.color_palette, 2)).sum(-1)
        dist_loss = (1 - dists / dists.max()).mean()
        valid_loss = (torch.floor(self.color_palette) * self.color_palette).sum()
        return valid_loss * params.palette_loss_valid + dist_loss * params.palette_loss_distinct

    def offset_loss(self, pred_offsets, params):
        return torch.pow(pred_offsets, 2).sum() * params.offset_loss

    def get_params(self, lr):
        return [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.weight_net.parameters(), "lr": lr},
                {"params": self.offset_net.parameters(), "lr": lr}, {"params": [self.color_palette], "lr": 2 * lr}]
