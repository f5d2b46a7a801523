"""Minimal driver of the occupancy-grid renderer: the call sequence of the reference's
NeRFRenderer.run_cuda / run_cuda_distill (nerf/renderer.py:259-392, 394-480) on the HIP operators.

It exists so the repo's own tests and bench.py can exercise the hot path end to end without the
reference checkout; with the reference present, its unmodified nerf/renderer.py drives the same
operators instead.  State kept like the reference: aabb_train/aabb_infer,
density_grid [C,128^3] fp32 (Morton order), density_bitfield [C*128^3/8] uint8, step_counter [16,2] int32 ring,
mean_density, iter_density, mean_count, local_step.

Also here: `run` (uniform sampling + optional importance resampling, renderer.py:128-256, `sample_pdf` :12-46) and
the occupancy-grid maintenance `mark_untrained_grid` / `update_extra_state` (renderer.py:482-649) on the
density-grid kernels of csrc/densitygrid.hip.
"""
import math

import torch
import torch.nn as nn

from . import raymarching


def sample_pdf(bins, weights, n_samples, det=False):
    """inverse-CDF resampling of NeRF (renderer.py:12-46): bins [B,T], weights [B,T-1] -> [B,n_samples]"""
    pdf = weights + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)                 # [B,T]
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=weights.device)
        u = u.expand(*cdf.shape[:-1], n_samples)
    else:
        u = torch.rand(*cdf.shape[:-1], n_samples, device=weights.device)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_lo + (u - cdf_lo) / denom * (bin_hi - bin_lo)


class NeRFRenderer(nn.Module):
    def __init__(self, model, bound=1, min_near=0.2, density_scale=1, grid_size=128, density_thresh=0.01,
                 filter_close_point=False):
        super().__init__()
        self.model = model
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))         # renderer.py:74
        self.grid_size = grid_size
        self.min_near = min_near
        self.density_scale = density_scale
        aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        self.density_thresh = density_thresh
        self.filter_close_point = filter_close_point
        self.register_buffer("density_grid", torch.zeros(self.cascade, grid_size ** 3))              # renderer.py:92-94
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * grid_size ** 3 // 8, dtype=torch.uint8))
        self.register_buffer("_grid_tmp", torch.zeros(grid_size ** 3, dtype=torch.int32), persistent=False)
        self.mean_density = 0
        self.iter_density = 0
        self.fused_post_ops = True
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        self.mean_count = 0
        self.local_step = 0

    # ---- checkpoints of the reference: there NeRFNetwork IS the renderer (class NeRFNetwork(NeRFRenderer)), so its
    # state_dict has `encoder.embeddings`, `sigma_net.weights`, ... next to `density_grid`, `aabb_train`, ...; here the network
    # is the attribute `model`.  load_state_dict accepts both layouts; state_dict(reference_layout=True) writes the reference's.
    def _to_local_keys(self, sd):
        own = {k for k, _ in self.named_buffers(recurse=False)} | {k for k, _ in self.named_parameters(recurse=False)}
        out = {}
        for k, v in sd.items():
            if k.startswith("model.") or k.split(".")[0] in own:
                out[k] = v
            else:
                out["model." + k] = v
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if "model" in state_dict and isinstance(state_dict["model"], dict):       # Trainer checkpoint (nerf/utils.py:1587): {'model': ...}
            state_dict = state_dict["model"]
        res = super().load_state_dict(self._to_local_keys(state_dict), strict=strict, assign=assign)
        for m in self.modules():                                                  # fp16 shadow tables follow the loaded parameters
            sh = getattr(m, "shadow", None)
            if sh is not None and hasattr(sh, "half"):
                p = getattr(m, "embeddings", None) if hasattr(m, "embeddings") else getattr(m, "weights", None)
                if p is not None:
                    sh.half.copy_(p.detach()); sh.version = p._version
        return res

    def state_dict(self, *args, reference_layout=False, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        if not reference_layout:
            return sd
        return type(sd)((k[len("model."):] if k.startswith("model.") else k, v) for k, v in sd.items())

    def reset_extra_state(self):
        """renderer.py:115-126"""
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    def density(self, x):
        return self.model.density(x)

    # ------------------------------------------------------------------ occupancy-grid maintenance
    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic):
        """cells no training camera covers get density -1 and are never sampled (renderer.py:482-554).
        poses [B,4,4] camera-to-world, intrinsic (fx, fy, cx, cy).  One kernel, poses staged through LDS."""
        raymarching.mark_untrained_grid(self.density_grid, poses, intrinsic, self.bound, self.min_near,
                                        self.filter_close_point, self.grid_size)
        return int((self.density_grid < 0).sum().item())

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, rng=None):
        """EMA-max refresh of the density grid + bitfield + mean_count (renderer.py:555-649).

        First 16 calls: every cell of every cascade is queried at a jittered position; afterwards H^3/4 uniformly
        random cells plus as many draws from the occupied cells.  `rng` (tests) replaces torch's generators:
        an object with rand(n, 3) -> [n,3] in [0,1) and randint(high, shape) -> int64.
        Per cascade: positions kernel -> grid encode -> fused sigma head -> scatter-max + EMA kernels.
        """
        H, dev = self.grid_size, self.density_grid.device
        rand = (lambda n: torch.rand(n, 3, device=dev)) if rng is None else (lambda n: rng.rand(n, 3).to(dev))
        randint = (lambda high, shape: torch.randint(0, high, shape, device=dev)) if rng is None else \
            (lambda high, shape: rng.randint(high, shape).to(dev))
        was_training = self.model.training
        self.model.eval()
        try:
            for cas in range(self.cascade):
                bound_c = min(2 ** cas, self.bound)
                if self.iter_density < 16:
                    n = H ** 3
                    if rng is None and H & (H - 1) == 0:
                        # every cell once, jitter i.i.d.: the order is free -- Morton order (point j = cell with Morton index j)
                        # instead of the reference's meshgrid order keeps neighbouring points in neighbouring cells
                        xyzs, indices = raymarching.density_grid_positions(
                            n, H, bound_c, noise=rand(n), coords=raymarching.morton3D_invert(torch.arange(n, device=dev, dtype=torch.int32)))
                    else:
                        xyzs, indices = raymarching.density_grid_positions(n, H, bound_c, noise=rand(n))
                elif rng is None:
                    # partial sweep, selection on the device: the occupied cells are counted, listed and drawn from without
                    # `nonzero`'s host read (round 3: 0.78 ms for half the points of the 0.82 ms full sweep).  The order of the
                    # points does not matter to the result (scatter-max + EMA), so both halves are drawn SORTED inside the call
                    # (order statistics, no sort; csrc/densitygrid.hip): two torch launches (the uniforms) instead of ~30.
                    n = H ** 3 // 4
                    if H & (H - 1) == 0:
                        xyzs, indices = raymarching.density_grid_partial_positions(self.density_grid[cas], None, None, H, bound_c,
                                                                                   noise=torch.rand(2 * n, 3, device=dev),
                                                                                   rnd=torch.rand(2, n + 1, device=dev), n=n)
                    else:
                        xyzs, indices = raymarching.density_grid_partial_positions(
                            self.density_grid[cas], torch.randint(0, H, (n, 3), device=dev, dtype=torch.int32), torch.rand(n, device=dev), H, bound_c,
                            noise=torch.rand(2 * n, 3, device=dev))
                else:                                            # recorded draws (tests): the reference's statements, one by one
                    n = H ** 3 // 4
                    coords = randint(H, (n, 3)).int()
                    occ = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                    if occ.numel() > 0:
                        occ = occ[randint(occ.shape[0], (n,))]
                        coords = torch.cat([coords, raymarching.morton3D_invert(occ.int())], dim=0)
                    xyzs, indices = raymarching.density_grid_positions(coords.shape[0], H, bound_c,
                                                                       noise=rand(coords.shape[0]), coords=coords)
                if hasattr(self.model, "density_sigma"):                      # sigma alone: no feature transpose, no geo_feat store
                    sigmas = self.model.density_sigma(xyzs).reshape(-1).float()
                else:
                    sigmas = self.density(xyzs)["sigma"].reshape(-1).float()  # caller's autocast state, like the reference
                raymarching.density_grid_update(self.density_grid[cas], sigmas, indices, self._grid_tmp,
                                                self.density_scale, decay)
        finally:
            self.model.train(was_training)
        # mean_density and the mean_count refresh (renderer.py:638, 644-647) need one host read each in the reference; here both
        # scalars travel in ONE device-to-host copy (the second read cost a second wait for an idle device)
        total_step = min(16, self.local_step)
        mean_t = torch.mean(self.density_grid.clamp(min=0)).double()
        cnt_t = self.step_counter[:total_step, 0].sum().double() if total_step > 0 else mean_t.new_zeros(())
        mean_v, cnt_v = torch.stack([mean_t, cnt_t]).tolist()
        self.mean_density = mean_v                                             # the fp32 mean, exactly what `.item()` returned
        self.iter_density += 1
        density_thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)
        if total_step > 0:
            self.mean_count = int(int(cnt_v) / total_step)
        self.local_step = 0

    def update_mean_count(self):
        """the mean_count refresh of update_extra_state (renderer.py:644-647); one D2H read every 16 steps"""
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    # ------------------------------------------------------------------ uniform-sampling render (renderer.py:128-256)
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False):
        """the reference's non-occupancy-grid path: `num_steps` uniform samples in [near, far] (+ `upsample_steps`
        importance samples), density for all, colour only where the weight exceeds 1e-4, cumprod compositing."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        sample_dist = (fars - nears) / num_steps
        z_vals = nears + (fars - nears) * torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0)   # [N,T]
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist

        def points(z):
            p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
            return torch.min(torch.max(p, aabb[:3]), aabb[3:])

        def weights_of(z, sigma):
            deltas = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], dim=-1)
            alphas = 1 - torch.exp(-deltas * self.density_scale * sigma)
            trans = torch.cumprod(torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1), dim=-1)
            return alphas * trans[..., :-1], deltas

        xyzs = points(z_vals)
        dens = {k: v.view(N, num_steps, -1) for k, v in self.density(xyzs.reshape(-1, 3)).items()}
        if upsample_steps > 0:
            with torch.no_grad():
                w, deltas = weights_of(z_vals, dens["sigma"].squeeze(-1))
                z_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z = sample_pdf(z_mid, w[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = points(new_z)
            new_dens = {k: v.view(N, upsample_steps, -1) for k, v in self.density(new_xyzs.reshape(-1, 3)).items()}
            z_vals, order = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], dim=1)
            xyzs = torch.gather(xyzs, 1, order.unsqueeze(-1).expand_as(xyzs))
            for k in dens:
                both = torch.cat([dens[k], new_dens[k]], dim=1)
                dens[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))
        weights, _ = weights_of(z_vals, dens["sigma"].squeeze(-1))
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        flat = {k: v.reshape(-1, v.shape[-1]) for k, v in dens.items()}
        mask = weights > 1e-4
        rgbs = self.model.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **flat).view(N, -1, 3)
        weights_sum = weights.sum(dim=-1)
        depth = torch.sum(weights * ((z_vals - nears) / (fars - nears)).clamp(0, 1), dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        image = image + (1 - weights_sum).unsqueeze(-1) * (1 if bg_color is None else bg_color)
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "weights_sum": weights_sum}

    # ------------------------------------------------------------------ training render (renderer.py:285-334)
    def render_train(self, rays_o, rays_d, bg_color=1, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024,
                     T_thresh=1e-4, dens_grid=None, gt=None, scaler=None):
        marched = self.march_train(rays_o, rays_d, perturb, force_all_rays, dt_gamma, max_steps, dens_grid)
        return self.shade_train(marched, bg_color, T_thresh, gt, scaler)

    def march_train(self, rays_o, rays_d, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024, dens_grid=None,
                    plan_backward=False):
        """first half of the training render: ray/box intersection and the occupancy march (renderer.py:285-311).  It
        reads no network weight, so it can run for step k+1 while step k is still in its backward (bench.py replays it
        as its own graph on a side stream; a forked branch inside ONE captured graph did not overlap on this stack).
        plan_backward=True additionally runs the position-only half of the hash-grid backward (counting pass + scans)
        here and returns the plan as a 7th element for shade_train."""
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        grid = self.density_bitfield if dens_grid is None else dens_grid
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
        counter = self.step_counter[self.local_step % 16]
        counter.zero_()
        self.local_step += 1
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, grid, self.cascade,
                                                                self.grid_size, nears, fars, counter, self.mean_count,
                                                                perturb, 128, force_all_rays, dt_gamma, max_steps)
        if plan_backward and hasattr(self.model, "plan_backward"):
            return xyzs, dirs, deltas, rays, nears, fars, self.model.plan_backward(xyzs)
        return xyzs, dirs, deltas, rays, nears, fars

    def shade_train(self, marched, bg_color=1, T_thresh=1e-4, gt=None, scaler=None):
        """second half: network on the samples, compositing, background blend, depth normalisation (renderer.py:313-334).
        gt [N,3] (optional, MI355X-native): also evaluate the trainer's criterion MSE(image, gt) (scaled by `scaler`'s
        loss scale) inside the compositing op -> result["loss"]; call loss.backward() on it."""
        xyzs, dirs, deltas, rays, nears, fars = marched[:6]
        plan = marched[6] if len(marched) > 6 else None
        sigmas, rgbs = self.model(xyzs, dirs, plan=plan) if plan is not None else self.model(xyzs, dirs)
        if self.density_scale != 1:
            sigmas = self.density_scale * sigmas
        loss = None
        if self.fused_post_ops and gt is not None:
            loss, weights_sum, depth, image = raymarching.composite_rays_train_blend_mse(sigmas, rgbs, deltas, rays, nears, fars,
                                                                                        gt, bg_color, T_thresh, scaler)
        elif self.fused_post_ops:    # composite + bg blend + depth normalisation in one kernel, gradients without zero fills
            weights_sum, depth, image = raymarching.composite_rays_train_blend(sigmas, rgbs, deltas, rays, nears, fars,
                                                                               bg_color, T_thresh)
        else:                        # operator-by-operator, as renderer.py:318-325
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        res = {"image": image, "depth": depth, "weights_sum": weights_sum, "nears": nears, "n_samples": xyzs.shape[0]}
        if loss is not None:
            res["loss"] = loss
        return res

    # ------------------------------------------------------------------ inference render (renderer.py:335-387)
    def _frame_loop_ok(self, rays_o):
        """the device-resident frame loop needs the default architecture (fused field) and fp16 evaluation"""
        m = self.model
        return (getattr(m, "fused_field", False) and rays_o.is_cuda and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.half)

    def _render_frame(self, rays_o, rays_d, grid, edit_bitfield, bg_color, perturb, dt_gamma, max_steps, T_thresh, scale_depth,
                      want_stats=False, row_budget=0, max_n_step=8):
        m = self.model
        enc, sn, cn = m.encoder, m.sigma_net, m.color_net
        table = enc.shadow.table_half(enc.embeddings) if enc.shadow is not None else enc.embeddings.detach().to(torch.half)
        ws = sn.shadow.table_half(sn.weights) if sn.shadow is not None else sn.weights.detach().half().contiguous()
        wc = cn.shadow.table_half(cn.weights) if cn.shadow is not None else cn.weights.detach().half().contiguous()
        noises = torch.rand(rays_o.shape[0], dtype=torch.float32, device=rays_o.device) if perturb else None
        return raymarching.render_frame(rays_o, rays_d, self.aabb_infer, self.min_near, grid, self.bound, self.cascade,
                                        self.grid_size, table, enc.offsets, enc.per_level_scale, enc.base_resolution, ws, wc,
                                        edit_bitfield=edit_bitfield, gridtype_id=enc.gridtype_id, align_corners=enc.align_corners,
                                        interp_id=enc.interp_id, density_scale=self.density_scale, dt_gamma=dt_gamma,
                                        max_steps=max_steps, T_thresh=T_thresh, max_n_step=max_n_step, row_budget=row_budget,
                                        noises=noises,
                                        bg_color=bg_color, scale_depth=scale_depth, want_stats=want_stats,
                                        offsets_host=getattr(enc, "offsets_host", None))

    _tile_perm_cache = {}

    @classmethod
    def _tile_perm(cls, H, W, th, tw, device):
        """ray order that walks the image in th x tw pixel tiles (rows of tiles, scanline inside a tile) and its inverse"""
        key = (H, W, th, tw, str(device))
        if key not in cls._tile_perm_cache:
            idx = torch.arange(H * W, device=device).view(H // th, th, W // tw, tw).permute(0, 2, 1, 3).reshape(-1)
            inv = torch.empty_like(idx)
            inv[idx] = torch.arange(H * W, device=device)
            cls._tile_perm_cache = {key: (idx, inv)}           # one entry: frames of one size come in a row
        return cls._tile_perm_cache[key]

    @torch.no_grad()
    def render_eval(self, rays_o, rays_d, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4,
                    scale_depth=True, dens_grid=None, device_compaction=True, frame_loop=True, want_stats=False, row_budget=0,
                    image_hw=None, tile_hw=(8, 4), max_n_step=8):
        if image_hw is not None and rays_o.numel() == 3 * image_hw[0] * image_hw[1] and image_hw[0] % tile_hw[0] == 0 and \
                image_hw[1] % tile_hw[1] == 0 and not (torch.is_tensor(bg_color) and bg_color.numel() > 3):
            # the rays are the pixels of one H x W image in scanline order: render them tile by tile (neighbouring rays
            # march through neighbouring cells, so a wave of the encoder touches fewer cache lines: 16.4 -> 15.6 ms on the
            # 800x800 bench frame) and hand the per-ray results back in the caller's order
            idx, inv = self._tile_perm(image_hw[0], image_hw[1], tile_hw[0], tile_hw[1], rays_o.device)
            res = self.render_eval(rays_o.reshape(-1, 3)[idx], rays_d.reshape(-1, 3)[idx], bg_color, perturb, dt_gamma, max_steps,
                                   T_thresh, scale_depth, dens_grid, device_compaction, frame_loop, want_stats, row_budget,
                                   max_n_step=max_n_step)
            return {k: (v[inv] if torch.is_tensor(v) and v.shape[:1] == inv.shape else v) for k, v in res.items()}
        """frame_loop=True (default, when the model is the default architecture under fp16 autocast): the whole loop runs
        as ONE backend call with its state on the device (lae_render_frame).  frame_loop=False: the reference's loop,
        operator by operator, with one host read of n_alive per iteration.  row_budget (both loops; the same schedule, so the
        same bits: tests/test_gpu_frame_fuzz.py): rows per iteration, 0 = N as in the reference (`n_step = max(min(N // n_alive, 8),
        1)`, renderer.py:363).  A larger budget keeps every ray's sample sequence up to the rounding of rays_t EXCEPT (a) for rays
        still alive at the `step < max_steps` cutoff (what a survivor has received by then depends on the schedule, in the reference
        as well) and (b) with perturb=True: the jitter moves only the samples of the FIRST march call and is dropped from rays_t
        afterwards (`last_t = t` after the jitter, raymarching.cu:747-749), so how many samples carry it is that call's n_step -- 1
        under the reference's rule, up to max_n_step under a boosted budget.  max_n_step: the 8 of
        that rule; with 1 every iteration takes ONE sample per alive ray, so a ray's result no longer depends on how many
        other rays share the call (the reference's schedule ties `rays_t` rounding to N: a sharded frame differs from the
        whole one by <= 1e-5; with max_n_step=1 the two are the same bits, tests/test_gpu_frame1080.py)."""
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        grid = self.density_bitfield if dens_grid is None else dens_grid
        N, device = rays_o.shape[0], rays_o.device
        if frame_loop and self._frame_loop_ok(rays_o):
            return self._render_frame(rays_o, rays_d, grid, None, bg_color, perturb, dt_gamma, max_steps, T_thresh, scale_depth,
                                      want_stats, row_budget, max_n_step)
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
        weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
        depth = torch.zeros(N, dtype=torch.float32, device=device)
        image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        n_alive = N
        rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        step = 0
        iterations = rows = 0
        budget = max(N, int(row_budget))                                             # rows per iteration; the reference: N
        while step < max_steps and n_alive > 0:
            n_step = max(min(budget // n_alive, max_n_step), 1)                      # renderer.py:363 (budget = N, max_n_step = 8)
            iterations += 1; rows += n_alive * n_step
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                        grid, self.cascade, self.grid_size, nears, fars, 128,
                                                        perturb if step == 0 else False, dt_gamma, max_steps)
            sigmas, rgbs = self.model(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth,
                                       image, T_thresh)
            if device_compaction:
                rays_alive, n_out = raymarching.compact_rays_alive(rays_alive, n_alive)
                n_alive = int(n_out.item())
            else:
                rays_alive = rays_alive[rays_alive >= 0]                             # renderer.py:375
                n_alive = rays_alive.shape[0]
            step += n_step
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        if scale_depth:
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        out = {"image": image, "depth": depth, "weights_sum": weights_sum}
        if want_stats:
            # alive_at_end > 0: the loop ended at the `step < max_steps` cutoff with rays still marching.  What such a ray has
            # received by then (sum of the iterations' n_step, between max_steps and max_steps + 7) depends on the SCHEDULE --
            # in the reference too (n_step follows N // n_alive) -- so only then does a boosted row budget change more than the
            # rounding of rays_t (tests/test_gpu_frame_fuzz.py)
            out["stats"] = {"iterations": iterations, "rows": rows, "steps": step, "alive_at_end": n_alive}
        return out

    # ------------------------------------------------------------------ distillation render (renderer.py:394-480)
    @torch.no_grad()
    def render_distill(self, rays_o, rays_d, edit_bitfield, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4,
                       grow_grid=False, frame_loop=True, perturb_depth=False, nears=None):
        """run_cuda_distill (renderer.py:394-480).  `nears` (optional): the rays' near bounds when the caller already has
        them (get_rays(aabb=...)); otherwise one near_far_from_aabb call provides `min_near` = nears.min() (:478)."""
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        dens = edit_bitfield if grow_grid else self.density_bitfield

        def finish(depth):                                                   # :466-469, :478
            if perturb_depth:
                depth = depth + (torch.rand(depth.shape, device=device) - 0.5) * (depth.max() - depth.min()) / max_steps
            return depth, rays_o + depth[..., None] * rays_d

        if frame_loop and self._frame_loop_ok(rays_o):
            res = self._render_frame(rays_o, rays_d, dens, edit_bitfield, None, perturb, dt_gamma, max_steps, T_thresh, False)
            if nears is None:
                nears, _ = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
            depth, x_term = finish(res["depth"])
            return {"image": res["image"], "depth": depth, "depth_edit": res["depth_edit"], "x_term": x_term,
                    "weights_edit": res["weights_edit"], "weights": res["weights_sum"], "min_near": nears.min()}
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
        weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
        weights_edit_sum = torch.zeros(N, dtype=torch.float32, device=device)
        depth = torch.zeros(N, dtype=torch.float32, device=device)
        depth_edit = torch.zeros(N, dtype=torch.float32, device=device)
        image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        n_alive = N
        rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        step = 0
        while step < max_steps and n_alive > 0:
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas, edit_occ = raymarching.march_rays_distill(
                n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, dens, edit_bitfield, self.cascade,
                self.grid_size, nears, fars, 128, perturb if step == 0 else False, dt_gamma, max_steps)
            sigmas, rgbs = self.model(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            raymarching.composite_rays_distill(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum,
                                               weights_edit_sum, depth, depth_edit, image, edit_occ, T_thresh)
            rays_alive, n_out = raymarching.compact_rays_alive(rays_alive, n_alive)
            n_alive = int(n_out.item())
            step += n_step
        depth, x_term = finish(depth)
        return {"image": image, "depth": depth, "depth_edit": depth_edit, "x_term": x_term,
                "weights_edit": weights_edit_sum, "weights": weights_sum, "min_near": nears.min()}
