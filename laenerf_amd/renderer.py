"""Minimal driver of the occupancy-grid renderer: the call sequence of the reference's
NeRFRenderer.run_cuda / run_cuda_distill (nerf/renderer.py:259-392, 394-480) on the HIP operators.

It exists so the repo's own tests and bench.py can exercise the hot path end to end without the
reference checkout; with the reference present, its unmodified nerf/renderer.py drives the same
operators instead.  State kept like the reference: aabb_train/aabb_infer,
density_bitfield [C*128^3/8] uint8, step_counter [16,2] int32 ring, mean_count, local_step.
"""
import math

import torch
import torch.nn as nn

from . import raymarching


class NeRFRenderer(nn.Module):
    def __init__(self, model, bound=1, min_near=0.2, density_scale=1, grid_size=128):
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
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * grid_size ** 3 // 8, dtype=torch.uint8))
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        self.mean_count = 0
        self.local_step = 0

    def update_mean_count(self):
        """the mean_count refresh of update_extra_state (renderer.py:644-647); one D2H read every 16 steps"""
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    # ------------------------------------------------------------------ training render (renderer.py:285-334)
    def render_train(self, rays_o, rays_d, bg_color=1, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024,
                     T_thresh=1e-4, dens_grid=None):
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
        sigmas, rgbs = self.model(xyzs, dirs)
        sigmas = self.density_scale * sigmas
        weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return {"image": image, "depth": depth, "weights_sum": weights_sum, "nears": nears, "n_samples": xyzs.shape[0]}

    # ------------------------------------------------------------------ inference render (renderer.py:335-387)
    @torch.no_grad()
    def render_eval(self, rays_o, rays_d, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4,
                    scale_depth=True, dens_grid=None, device_compaction=True):
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        grid = self.density_bitfield if dens_grid is None else dens_grid
        N, device = rays_o.shape[0], rays_o.device
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
        weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
        depth = torch.zeros(N, dtype=torch.float32, device=device)
        image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        n_alive = N
        rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        step = 0
        while step < max_steps and n_alive > 0:
            n_step = max(min(N // n_alive, 8), 1)                                    # renderer.py:363
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
        return {"image": image, "depth": depth, "weights_sum": weights_sum}

    # ------------------------------------------------------------------ distillation render (renderer.py:394-480)
    @torch.no_grad()
    def render_distill(self, rays_o, rays_d, edit_bitfield, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4,
                       grow_grid=False):
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
        dens = edit_bitfield if grow_grid else self.density_bitfield
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
        x_term = rays_o + depth[..., None] * rays_d
        return {"image": image, "depth": depth, "depth_edit": depth_edit, "x_term": x_term,
                "weights_edit": weights_edit_sum, "weights": weights_sum}
