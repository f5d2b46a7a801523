"""Tensor-level backends with the exact function names / argument order of the reference's
four pybind11 modules, implemented on the C ABI (include/laenerf.h).

    _raymarching  raymarching/src/bindings.cpp:5-20
    _gridencoder  gridencoder/src/bindings.cpp:5-8
    _shencoder    shencoder/src/bindings.cpp:5-7
    _ffmlp        ffmlp/src/bindings.cpp:5-10

`install_as_reference_backends()` registers them in sys.modules under those names, which
is what the reference's wrappers import first (`try: import _raymarching as _backend`,
raymarching/raymarching.py:9-12), so nerf/renderer.py and editing/* run unmodified.
"""
import sys
import types

import numpy as np
import torch

from . import _lib
from ._lib import check, need_contig, need_cuda, ptr, stream

_F16 = torch.float16


def _dtype_code(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.float16:
        return 1
    raise RuntimeError(f"laenerf_amd: unsupported dtype {t.dtype} (float32 / float16 only)")


def _host_i32(a, n):
    """host pointer of a C-contiguous int32 numpy array with n entries (None -> NULL); the array must outlive the call only"""
    if a is None:
        return None
    if not (isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags.c_contiguous and a.size == n):
        raise RuntimeError("laenerf_amd: offsets_host must be a contiguous int32 numpy array with L + 1 entries")
    return a.ctypes.data


def _need_f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("laenerf_amd.raymarching: float32 tensors required (the reference's wrappers cast to fp32)")


_scratch = {}
_scratch_retired = []


def _workspace(device, nbytes):
    """grow-only per-device scratch buffer for scans / compaction (stream-ordered reuse).  Same policy as the library's
    own workspaces (include/laenerf.h): an outgrown buffer is retired, not freed -- captured graphs and queued kernels
    keep the address they were given -- and growth inside a stream capture is refused (warm up eagerly first)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("laenerf_amd: scratch buffer must grow inside a stream capture; run the same call once "
                               "eagerly (warm-up) at the largest size before capturing")
        if buf is not None:
            _scratch_retired.append(buf)
        buf = torch.empty(max(int(nbytes) * 3 // 2, 1 << 20), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def free_workspaces():
    """drop every scratch buffer (Python side and library side).  Only when no captured graph that used the backend
    will be replayed again."""
    torch.cuda.synchronize()
    _scratch.clear()
    _scratch_retired.clear()
    check(_lib.load().lae_free_workspaces(), "free_workspaces")


# --------------------------------------------------------------------------- _raymarching
# Loss values whose final sum was deferred (composite_rays_train_step(defer_loss=True)).  Round 5: keyed by
# (device, data_ptr of the criterion node's grad_sigmas buffer) -- the tensor that autograd hands from the criterion node's
# backward to the fused head's backward, i.e. the edge between exactly those two autograd nodes -- instead of one slot per
# process: LAENeRF's flow holds two models, two optimizers and one scaler in one process (nerf/utils.py:969-972, 1041-1043), and
# with one slot whichever head backward ran next summed whatever value was pending, also another model's on another stream.
# Entry: (partials, n_part, n_elem, scale, loss_out, stream handle of the forward, made inside a capture?).  The head backward
# that consumes THIS gradient buffer on THIS stream takes the value along in its reduction launch; everything else is finished
# by flush_pending_loss() (a launch of its own, on the stream the value was made on).
_pending_loss = {}
_PENDING_MAX = 8                      # forwards whose backward never ran (no_grad loops) do not pile up: the oldest is finished
deferred_loss_stats = {"carried": 0, "flushed": 0}


def _loss_key(grad_sigmas):
    dev = grad_sigmas.device
    return (dev.index if dev.index is not None else torch.cuda.current_device(), grad_sigmas.data_ptr())


def _finish_entry(entry):
    partials, n_part, n_elem, scale, loss_out, strm, _ = entry
    with torch.cuda.device(loss_out.device):
        check(_lib.load().lae_loss_finish(ptr(partials), n_part, n_elem, ptr(scale), ptr(loss_out), strm), "loss_finish")
    deferred_loss_stats["flushed"] += 1


def flush_pending_loss(device=None):
    """finish every deferred loss value (of `device`, default all) that no head backward took along"""
    for key in [k for k in _pending_loss if device is None or k[0] == device]:
        _finish_entry(_pending_loss.pop(key))


def retarget_pending_loss(old_grad_sigmas, new_grad_sigmas):
    """the criterion node scaled its stored gradients by a general upstream gradient: the value now rides with the new tensor"""
    e = _pending_loss.pop(_loss_key(old_grad_sigmas), None)
    if e is not None:
        _pending_loss[_loss_key(new_grad_sigmas)] = e


class _RayMarching:
    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        need_cuda(rays_o, rays_d, aabb, nears, fars); need_contig(rays_o, rays_d, aabb, nears, fars)
        _need_f32(rays_o, rays_d, aabb, nears, fars)
        check(_lib.load().lae_near_far_from_aabb(ptr(rays_o), ptr(rays_d), ptr(aabb), N, min_near, ptr(nears), ptr(fars),
                                                 stream()), "near_far_from_aabb")

    @staticmethod
    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        need_cuda(rays_o, rays_d, coords); need_contig(rays_o, rays_d, coords); _need_f32(rays_o, rays_d, coords)
        check(_lib.load().lae_sph_from_ray(ptr(rays_o), ptr(rays_d), radius, N, ptr(coords), stream()), "sph_from_ray")

    @staticmethod
    def morton3D(coords, N, indices):
        need_cuda(coords, indices); need_contig(coords, indices)
        assert coords.dtype == torch.int32 and indices.dtype == torch.int32
        check(_lib.load().lae_morton3D(ptr(coords), N, ptr(indices), stream()), "morton3D")

    @staticmethod
    def morton3D_invert(indices, N, coords):
        need_cuda(coords, indices); need_contig(coords, indices)
        assert coords.dtype == torch.int32 and indices.dtype == torch.int32
        check(_lib.load().lae_morton3D_invert(ptr(indices), N, ptr(coords), stream()), "morton3D_invert")

    @staticmethod
    def packbits(grid, N, density_thresh, bitfield):
        need_cuda(grid, bitfield); need_contig(grid, bitfield); _need_f32(grid)
        assert bitfield.dtype == torch.uint8
        check(_lib.load().lae_packbits(ptr(grid), N, float(density_thresh), ptr(bitfield), stream()), "packbits")

    # ---- occupancy-grid maintenance (Python in the reference, nerf/renderer.py:482-649; kernels here)
    @staticmethod
    def density_grid_positions(coords, n, H, bound_c, noise, xyzs, indices):
        need_cuda(coords, noise, xyzs, indices); need_contig(coords, noise, xyzs, indices); _need_f32(noise, xyzs)
        assert indices.dtype == torch.int32 and (coords is None or coords.dtype == torch.int32)
        assert xyzs.numel() == 3 * n and indices.numel() == n and (noise is None or noise.numel() == 3 * n)
        check(_lib.load().lae_density_grid_positions(ptr(coords), n, H, float(bound_c), ptr(noise), ptr(xyzs), ptr(indices),
                                                     stream()), "density_grid_positions")

    @staticmethod
    def density_grid_partial_positions(grid_c, coords_rand, u, rnd, n, H, bound_c, noise, xyzs, indices):
        """MI355X-native: the point selection of update_extra_state's partial sweep without a host read (include/laenerf.h);
        rnd [2, n+1]: sorted draws made on the device, else coords_rand [n,3] + u [n]"""
        ts = (grid_c, coords_rand, u, rnd, noise, xyzs, indices)
        need_cuda(*ts); need_contig(*ts); _need_f32(grid_c, u, rnd, noise, xyzs)
        assert indices.dtype == torch.int32 and (coords_rand is None or coords_rand.dtype == torch.int32)
        assert grid_c.numel() == H ** 3 and xyzs.numel() == 6 * n and indices.numel() == 2 * n and (noise is None or noise.numel() == 6 * n)
        assert (rnd is not None and rnd.numel() == 2 * (n + 1)) or (coords_rand.numel() == 3 * n and u.numel() == n)
        lib = _lib.load()
        ws = _workspace(grid_c.device, lib.lae_density_grid_partial_scratch_bytes(grid_c.numel(), n))
        check(lib.lae_density_grid_partial_positions(ptr(grid_c), grid_c.numel(), ptr(coords_rand), ptr(u), ptr(rnd), n, H, float(bound_c),
                                                     ptr(noise), ptr(xyzs), ptr(indices), ptr(ws), stream()), "density_grid_partial_positions")

    @staticmethod
    def density_grid_update(sigmas, indices, n, density_scale, decay, cells, grid, tmp):
        need_cuda(sigmas, indices, grid, tmp); need_contig(sigmas, indices, grid, tmp); _need_f32(sigmas, grid)
        assert indices.dtype == torch.int32 and tmp.dtype == torch.int32
        assert grid.numel() == cells and tmp.numel() == cells and sigmas.numel() >= n and indices.numel() >= n
        check(_lib.load().lae_density_grid_update(ptr(sigmas), ptr(indices), n, float(density_scale), float(decay), cells,
                                                  ptr(grid), ptr(tmp), stream()), "density_grid_update")

    @staticmethod
    def mark_untrained_grid(poses, B, fx, fy, cx, cy, C, H, bound, min_near, filter_close_point, grid):
        need_cuda(poses, grid); need_contig(poses, grid); _need_f32(poses, grid)
        assert poses.numel() == B * 16 and grid.numel() == C * H ** 3
        check(_lib.load().lae_mark_untrained_grid(ptr(poses), B, float(fx), float(fy), float(cx), float(cy), C, H,
                                                  float(bound), float(min_near), int(bool(filter_close_point)), ptr(grid),
                                                  stream()), "mark_untrained_grid")

    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                         rays, counter, noises, rows_end=None):
        """`rows_end` (optional int32[1], MI355X extension): receives the first sample row no ray owns; rows from there
        to M are zero-filled by the kernel, so xyzs / dirs / deltas may be allocated uninitialised"""
        need_cuda(rays_o, rays_d, grid, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
        need_contig(rays_o, rays_d, grid, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
        _need_f32(rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
        assert grid.dtype == torch.uint8 and rays.dtype == torch.int32 and counter.dtype == torch.int32
        lib = _lib.load()
        ws = _workspace(rays_o.device, lib.lae_march_rays_train_scratch_bytes(N))
        check(lib.lae_march_rays_train(ptr(rays_o), ptr(rays_d), ptr(grid), bound, dt_gamma, max_steps, N, C, H, M,
                                       ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays),
                                       ptr(counter), ptr(noises), ptr(ws), ptr(rows_end), stream()), "march_rays_train")

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        need_cuda(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        need_contig(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        _need_f32(sigmas, rgbs, deltas, weights_sum, depth, image)
        check(_lib.load().lae_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, T_thresh,
                                                           ptr(weights_sum), ptr(depth), ptr(image), stream()),
              "composite_rays_train_forward")

    @staticmethod
    def composite_rays_train_forward_blend(sigmas, rgbs, deltas, rays, M, N, T_thresh, nears, fars, bg_rays, bg, weights_sum,
                                           depth, image, depth_out, image_out):
        ts = (sigmas, rgbs, deltas, rays, nears, fars, bg_rays, weights_sum, depth, image, depth_out, image_out)
        need_cuda(*ts); need_contig(*ts)
        _need_f32(sigmas, rgbs, deltas, nears, fars, bg_rays, weights_sum, depth, image, depth_out, image_out)
        check(_lib.load().lae_composite_rays_train_forward_blend(
            ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, T_thresh, ptr(nears), ptr(fars), ptr(bg_rays), bg[0], bg[1],
            bg[2], ptr(weights_sum), ptr(depth), ptr(image), ptr(depth_out), ptr(image_out), stream()),
            "composite_rays_train_forward_blend")

    @staticmethod
    def composite_rays_train_step(sigmas, rgbs, deltas, rays, M, N, T_thresh, nears, fars, bg_rays, bg, rows_end, target, scale,
                                  weights_sum, depth, image, depth_out, image_out, grad_image, grad_sigmas, grad_rgbs, loss_out,
                                  partials, defer_loss=False):
        """MI355X extension: compositing forward (+ blend) + MSE criterion + compositing backward in one launch.
        defer_loss: the loss VALUE (loss_out, NaN until then) is summed later -- by the fused head's backward, which takes it
        along in its reduction launch, or by flush_pending_loss() (laenerf_amd.optim.FusedAdam.backward / step call it)."""
        ts = (sigmas, rgbs, deltas, rays, nears, fars, bg_rays, rows_end, target, scale, weights_sum, depth, image, depth_out,
              image_out, grad_image, grad_sigmas, grad_rgbs, loss_out, partials)
        need_cuda(*ts); need_contig(*ts)
        _need_f32(sigmas, rgbs, deltas, nears, fars, bg_rays, target, scale, weights_sum, depth, image, depth_out, image_out,
                  grad_image, grad_sigmas, grad_rgbs, loss_out, partials)
        if partials.numel() < (N + 3) // 4:
            raise RuntimeError("composite_rays_train_step: partials needs cdiv(N, 4) floats")
        check(_lib.load().lae_composite_rays_train_step(
            ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, T_thresh, ptr(nears), ptr(fars), ptr(bg_rays), bg[0], bg[1], bg[2],
            ptr(rows_end), ptr(target), ptr(scale), ptr(weights_sum), ptr(depth), ptr(image), ptr(depth_out), ptr(image_out),
            ptr(grad_image), ptr(grad_sigmas), ptr(grad_rgbs), ptr(loss_out), ptr(partials), int(bool(defer_loss)), stream()),
            "composite_rays_train_step")
        if defer_loss and N > 0:                      # N == 0: the library returned at once, there is nothing to finish
            capturing = torch.cuda.is_current_stream_capturing()
            key = _loss_key(grad_sigmas)
            # entries that must be finished now: one under the same key (its gradient buffer was freed and reused: its backward
            # can never run) and the oldest ones beyond the cap
            stale = [k for k in _pending_loss if k == key]
            others = [k for k in _pending_loss if k != key]
            stale += others[:max(0, len(others) + 1 - _PENDING_MAX)]
            for k in stale:
                if capturing and not _pending_loss[k][6]:
                    # an older, eagerly made value would be finished INSIDE the capture: its buffers would be baked into the
                    # graph and freed with the entry; refuse instead
                    raise RuntimeError("composite_rays_train_step(defer_loss=True) inside a stream capture while an eager deferred loss "
                                       "is pending: call laenerf_amd.backend.flush_pending_loss() before capturing")
                _finish_entry(_pending_loss.pop(k))
            _pending_loss[key] = (partials, (N + 3) // 4, 3 * N, scale, loss_out, stream(), capturing)

    @staticmethod
    def composite_rays_train_backward_blend(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M,
                                            N, T_thresh, bg_rays, bg, rows_end, grad_sigmas, grad_rgbs, grad_scale=None):
        """grad_scale (MI355X extension): device scalar multiplied into the incoming gradients; grad_weights_sum may be None"""
        ts = (grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, bg_rays, rows_end, grad_sigmas,
              grad_rgbs, grad_scale)
        need_cuda(*ts); need_contig(*ts)
        _need_f32(grad_weights_sum, grad_image, sigmas, rgbs, deltas, weights_sum, image, bg_rays, grad_sigmas, grad_rgbs, grad_scale)
        check(_lib.load().lae_composite_rays_train_backward_blend_ex(
            ptr(grad_weights_sum), ptr(grad_image), ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), ptr(weights_sum),
            ptr(image), M, N, T_thresh, ptr(bg_rays), bg[0], bg[1], bg[2], ptr(rows_end), ptr(grad_scale), ptr(grad_sigmas),
            ptr(grad_rgbs), stream()), "composite_rays_train_backward_blend")

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                      T_thresh, grad_sigmas, grad_rgbs):
        ts = (grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, grad_sigmas, grad_rgbs)
        need_cuda(*ts); need_contig(*ts)
        _need_f32(grad_weights_sum, grad_image, sigmas, rgbs, deltas, weights_sum, image, grad_sigmas, grad_rgbs)
        check(_lib.load().lae_composite_rays_train_backward(ptr(grad_weights_sum), ptr(grad_image), ptr(sigmas), ptr(rgbs),
                                                            ptr(deltas), ptr(rays), ptr(weights_sum), ptr(image), M, N,
                                                            T_thresh, ptr(grad_sigmas), ptr(grad_rgbs), stream()),
              "composite_rays_train_backward")

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears,
                   fars, xyzs, dirs, deltas, noises):
        ts = (rays_alive, rays_t, rays_o, rays_d, grid, nears, fars, xyzs, dirs, deltas, noises)
        need_cuda(*ts); need_contig(*ts)
        _need_f32(rays_t, rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
        check(_lib.load().lae_march_rays(n_alive, n_step, ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), bound,
                                         dt_gamma, max_steps, C, H, ptr(grid), ptr(nears), ptr(fars), ptr(xyzs),
                                         ptr(dirs), ptr(deltas), ptr(noises), stream()), "march_rays")

    @staticmethod
    def march_rays_distill(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid,
                           edit_grid, nears, fars, xyzs, dirs, deltas, int_edit, noises):
        ts = (rays_alive, rays_t, rays_o, rays_d, grid, edit_grid, nears, fars, xyzs, dirs, deltas, int_edit, noises)
        need_cuda(*ts); need_contig(*ts)
        _need_f32(rays_t, rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
        assert int_edit.dtype in (torch.bool, torch.uint8)
        check(_lib.load().lae_march_rays_distill(n_alive, n_step, ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d),
                                                 bound, dt_gamma, max_steps, C, H, ptr(grid), ptr(edit_grid), ptr(nears),
                                                 ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(int_edit), ptr(noises),
                                                 stream()), "march_rays_distill")

    @staticmethod
    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights, depth, image):
        ts = (rays_alive, rays_t, sigmas, rgbs, deltas, weights, depth, image)
        need_cuda(*ts); need_contig(*ts); _need_f32(rays_t, sigmas, rgbs, deltas, weights, depth, image)
        check(_lib.load().lae_composite_rays(n_alive, n_step, T_thresh, ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs),
                                             ptr(deltas), ptr(weights), ptr(depth), ptr(image), stream()), "composite_rays")

    @staticmethod
    def composite_rays_distill(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights, weights_edit,
                               depth, depth_edit, int_edit, image):
        ts = (rays_alive, rays_t, sigmas, rgbs, deltas, weights, weights_edit, depth, depth_edit, int_edit, image)
        need_cuda(*ts); need_contig(*ts)
        _need_f32(rays_t, sigmas, rgbs, deltas, weights, weights_edit, depth, depth_edit, image)
        check(_lib.load().lae_composite_rays_distill(n_alive, n_step, T_thresh, ptr(rays_alive), ptr(rays_t), ptr(sigmas),
                                                     ptr(rgbs), ptr(deltas), ptr(weights), ptr(weights_edit), ptr(depth),
                                                     ptr(depth_edit), ptr(int_edit), ptr(image), stream()),
              "composite_rays_distill")

    # MI355X-native extension (no reference counterpart): the whole inference loop on the device
    @staticmethod
    def render_frame(rays_o, rays_d, N, aabb, min_near, grid, edit_grid, bound, dt_gamma, max_steps, C, H, table_f16, offsets, L, S,
                     base_resolution, gridtype, align_corners, interp, sigma_weights, color_weights, density_scale, T_thresh,
                     max_n_step, row_budget, noises, bg_rays, bg_rgb, blend_bg, scale_depth, weights_sum, depth, image,
                     weights_edit, depth_edit, want_stats=False, offsets_host=None):
        """whole inference loop of run_cuda / run_cuda_distill as one call (include/laenerf.h lae_render_frame)"""
        import ctypes
        tensors = (rays_o, rays_d, aabb, grid, edit_grid, table_f16, offsets, sigma_weights, color_weights, noises, bg_rays,
                   weights_sum, depth, image, weights_edit, depth_edit)
        need_cuda(*tensors); need_contig(*tensors)
        if rays_o.dtype != torch.float32 or rays_d.dtype != torch.float32 or table_f16.dtype != torch.half or \
                sigma_weights.dtype != torch.half or color_weights.dtype != torch.half or offsets.dtype != torch.int32 or \
                grid.dtype != torch.uint8:
            raise RuntimeError("render_frame: rays float32, table / MLP weights float16, offsets int32, bitfield uint8")
        lib = _lib.load()
        nbytes = lib.lae_render_frame_workspace_bytes(N, L, int(row_budget))
        key = ("frame", rays_o.device.index if rays_o.device.index is not None else torch.cuda.current_device())
        ws = _scratch.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is not None:
                _scratch_retired.append(ws)               # same policy as _workspace: never free what the stream may still use
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=rays_o.device)
            _scratch[key] = ws
        stats = (ctypes.c_uint32 * 4)() if want_stats else None
        check(lib.lae_render_frame(ptr(rays_o), ptr(rays_d), N, ptr(aabb), float(min_near), ptr(grid), ptr(edit_grid), float(bound),
                                   float(dt_gamma), max_steps, C, H, ptr(table_f16), ptr(offsets), _host_i32(offsets_host, L + 1), L, float(S),
                                   base_resolution,
                                   gridtype, int(bool(align_corners)), interp, ptr(sigma_weights), ptr(color_weights),
                                   float(density_scale), float(T_thresh), max_n_step, int(row_budget), ptr(noises), ptr(bg_rays), float(bg_rgb[0]),
                                   float(bg_rgb[1]), float(bg_rgb[2]), int(bool(blend_bg)), int(bool(scale_depth)), ptr(weights_sum),
                                   ptr(depth), ptr(image), ptr(weights_edit), ptr(depth_edit), ptr(ws), ws.numel(),
                                   ctypes.cast(stats, ctypes.c_void_p) if want_stats else None, stream()), "render_frame")
        if want_stats:
            # the call waited for the loop's last iteration: a time-out after that point can only be the tail's
            if lib.lae_render_frame_last_status() == 1:
                raise RuntimeError("laenerf_amd.render_frame: a cross-stream wait timed out behind the call; the frame's outputs are NaN")
            return {"iterations": stats[0], "rows": stats[1], "iterations_launched": stats[2]}
        return None

    @staticmethod
    def render_frame_set_overlap(on):
        """A/B switch: lookahead marcher on a side stream beside the network kernels (default) or in-line"""
        check(_lib.load().lae_render_frame_set_overlap(int(bool(on))), "render_frame_set_overlap")

    @staticmethod
    def render_frame_probe():
        """-> {"in_use": candidate index or -1, "handshake_us": [...]}: the frame loop's most recent side-stream probe (candidate 0 =
        highest priority, 1.. = the caller's class; -1.0 = not probed / not concurrent; include/laenerf.h lae_render_frame_probe_us)"""
        import ctypes
        us = (ctypes.c_float * 5)(*([-1.0] * 5))
        used = int(_lib.load().lae_render_frame_probe_us(ctypes.cast(us, ctypes.c_void_p), 5))
        return {"in_use": used, "handshake_us": [round(float(v), 1) for v in us]}

    @staticmethod
    def render_frame_mode():
        """1 while frames overlap their lookahead on the side stream; 0 once switched off, probed as not concurrent, or degraded
        after a cross-stream wait timed out (include/laenerf.h)"""
        return int(_lib.load().lae_render_frame_mode())

    @staticmethod
    def render_frame_last_status():
        """after synchronising the frame's stream: 0 = the most recent frame completed, 1 = a cross-stream wait of it timed out after
        the call had returned (its outputs are NaN: render again), -1 = no frame yet (include/laenerf.h)"""
        return int(_lib.load().lae_render_frame_last_status())

    @staticmethod
    def render_frame_check():
        """synchronise the current stream and raise if the most recent frame was poisoned by a timed-out wait"""
        torch.cuda.current_stream().synchronize()
        if _lib.load().lae_render_frame_last_status() == 1:
            raise RuntimeError("laenerf_amd.render_frame: a cross-stream wait of the most recent frame timed out after the call returned; "
                               "its outputs are NaN -- render the frame again (the library runs in line from now on)")

    # MI355X-native extension (no reference counterpart): device-side alive-list compaction
    @staticmethod
    def compact_rays_alive(rays_alive, n_alive, out_alive, n_out_dev):
        need_cuda(rays_alive, out_alive, n_out_dev)
        lib = _lib.load()
        ws = _workspace(rays_alive.device, lib.lae_compact_scratch_bytes(n_alive))
        check(lib.lae_compact_rays_alive(ptr(rays_alive), n_alive, ptr(out_alive), ptr(n_out_dev), ptr(ws), stream()),
              "compact_rays_alive")


# --------------------------------------------------------------------------- _gridencoder
class _GridEncoder:
    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                            interp, blc=False, in_map=(0.0, 1.0), offsets_host=None):
        """blc / in_map / offsets_host are MI355X extensions: [B, L*C] output layout; coordinates read as
        (x + in_map[0]) * in_map[1]; the module's host copy of `offsets` (int32 numpy array, include/laenerf.h)"""
        need_cuda(inputs, embeddings, offsets, outputs, dy_dx); need_contig(inputs, embeddings, offsets, outputs, dy_dx)
        if inputs.dtype != torch.float32 or offsets.dtype != torch.int32:
            raise RuntimeError("grid_encode_forward: inputs must be float32, offsets int32")   # gridencoder.cu:461-463
        if outputs.dtype != embeddings.dtype or (dy_dx is not None and dy_dx.dtype != embeddings.dtype):
            raise RuntimeError("grid_encode_forward: outputs/dy_dx must have the embeddings dtype")
        lib = _lib.load()
        args = (ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs), B, D, C, L, float(S), H, ptr(dy_dx), gridtype,
                int(bool(align_corners)), interp, _dtype_code(embeddings))
        if blc or tuple(in_map) != (0.0, 1.0) or offsets_host is not None:
            check(lib.lae_grid_encode_forward_ex(*args, int(bool(blc)), float(in_map[0]), float(in_map[1]), _host_i32(offsets_host, L + 1),
                                                 stream()), "grid_encode_forward")
        else:
            check(lib.lae_grid_encode_forward(*args, stream()), "grid_encode_forward")

    @staticmethod
    def grid_backward_plan(inputs, offsets, B, D, C, L, S, H, gridtype, align_corners, interp, half, in_map=(0.0, 1.0),
                           offsets_host=None, touched_lines=None):
        """first half of the binned grid backward (positions only: count pass + scans) -> plan tensor for
        grid_encode_backward(plan=...).  touched_lines: address of the "ever touched" bitmap this pass maintains (needs
        offsets_host); the returned tensor carries `.marks_touched` accordingly"""
        need_cuda(inputs, offsets); need_contig(inputs, offsets)
        lib = _lib.load()
        plan = torch.empty(int(lib.lae_grid_backward_plan_bytes(B, L)), dtype=torch.uint8, device=inputs.device)
        check(lib.lae_grid_encode_backward_plan(ptr(inputs), ptr(offsets), B, D, C, L, float(S), H, gridtype, int(bool(align_corners)),
                                                interp, 1 if half else 0, float(in_map[0]), float(in_map[1]),
                                                _host_i32(offsets_host, L + 1), ptr(plan), touched_lines, stream()),
              "grid_backward_plan")
        plan.marks_touched = touched_lines is not None
        return plan

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                             gridtype, align_corners, interp, blc=False, in_map=(0.0, 1.0), offsets_host=None, plan=None,
                             nonfinite_flag=None, touched_lines=None):
        """plan (MI355X extension): result of grid_backward_plan for the same inputs -- skips the count pass and scans.
        nonfinite_flag: address (int) of a device int32 that is OR-ed with 1 when the call stores a non-finite table gradient;
        touched_lines: address (int) of the "ever touched" bitmap, one bit per 8 table entries (include/laenerf.h: binned
        pipeline only)"""
        ts = (grad, inputs, embeddings, offsets, grad_embeddings, dy_dx, grad_inputs, plan)
        need_cuda(*ts); need_contig(*ts)
        if grad.dtype != grad_embeddings.dtype:
            raise RuntimeError("grid_encode_backward: grad and grad_embeddings dtypes differ")
        lib = _lib.load()
        if plan is not None:
            if blc or dy_dx is not None:
                raise RuntimeError("grid_encode_backward: a plan needs level-major gradients and no input gradient")
            if touched_lines is not None and not getattr(plan, "marks_touched", False):
                raise RuntimeError("grid_encode_backward: the plan was made without the touched-lines bitmap (pass it to grid_backward_plan)")
            check(lib.lae_grid_encode_backward_planned(ptr(grad), ptr(inputs), ptr(offsets), ptr(grad_embeddings), B, D, C, L, float(S), H,
                                                       gridtype, int(bool(align_corners)), interp, _dtype_code(grad), float(in_map[0]),
                                                       float(in_map[1]), _host_i32(offsets_host, L + 1), ptr(plan), nonfinite_flag, stream()),
                  "grid_encode_backward")
            return
        args = (ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), B, D, C, L, float(S), H,
                ptr(dy_dx), ptr(grad_inputs), gridtype, int(bool(align_corners)), interp, _dtype_code(grad))
        if blc or tuple(in_map) != (0.0, 1.0) or offsets_host is not None or nonfinite_flag is not None or touched_lines is not None:
            check(lib.lae_grid_encode_backward_ex(*args, int(bool(blc)), float(in_map[0]), float(in_map[1]), _host_i32(offsets_host, L + 1),
                                                  nonfinite_flag, touched_lines, stream()), "grid_encode_backward")
        else:
            check(lib.lae_grid_encode_backward(*args, stream()), "grid_encode_backward")

    @staticmethod
    def grid_backward_workspace_bytes(B, L, half=True):
        return int(_lib.load().lae_grid_backward_workspace_bytes(B, L, 1 if half else 0))

    @staticmethod
    def set_backward_mode(mode):
        """0 = binned LDS pipeline (default), 1 = generic global-atomic kernel"""
        check(_lib.load().lae_grid_set_backward_mode(int(mode)), "grid_set_backward_mode")

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
        need_cuda(inputs, embeddings, grad, offsets); need_contig(inputs, embeddings, grad, offsets)
        check(_lib.load().lae_grad_total_variation(ptr(inputs), ptr(embeddings), ptr(grad), ptr(offsets), float(weight), B, D,
                                                   C, L, float(S), H, gridtype, int(bool(align_corners)),
                                                   _dtype_code(embeddings), stream()), "grad_total_variation")


# --------------------------------------------------------------------------- _shencoder
class _SHEncoder:
    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
        need_cuda(inputs, outputs, dy_dx); need_contig(inputs, outputs, dy_dx)
        if inputs.dtype != torch.float32 or outputs.dtype != torch.float32:
            raise RuntimeError("sh_encode_forward: float32 required (sphere_harmonics.py:16 casts to fp32)")
        check(_lib.load().lae_sh_encode_forward(ptr(inputs), ptr(outputs), B, D, C, ptr(dy_dx), stream()), "sh_encode_forward")

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        need_cuda(grad, inputs, dy_dx, grad_inputs); need_contig(grad, inputs, dy_dx, grad_inputs)
        if grad.dtype != torch.float32:
            raise RuntimeError("sh_encode_backward: float32 required")
        check(_lib.load().lae_sh_encode_backward(ptr(grad), ptr(inputs), B, D, C, ptr(dy_dx), ptr(grad_inputs), stream()),
              "sh_encode_backward")


# --------------------------------------------------------------------------- _ffmlp
class _FreqEncoder:
    """freqencoder/src/bindings.cpp:5-8"""

    @staticmethod
    def freq_encode_forward(inputs, B, D, deg, C, outputs):
        need_cuda(inputs, outputs); need_contig(inputs, outputs); _need_f32(inputs, outputs)     # freqencoder.cu:98-104
        check(_lib.load().lae_freq_encode_forward(ptr(inputs), B, D, deg, C, ptr(outputs), stream()), "freq_encode_forward")

    @staticmethod
    def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
        need_cuda(grad, outputs, grad_inputs); need_contig(grad, outputs, grad_inputs); _need_f32(grad, outputs, grad_inputs)
        check(_lib.load().lae_freq_encode_backward(ptr(grad), ptr(outputs), B, D, deg, C, ptr(grad_inputs), stream()),
              "freq_encode_backward")


class _FFMLP:
    @staticmethod
    def _half(*ts):
        for t in ts:
            if t is not None and t.dtype != _F16:
                raise RuntimeError("ffmlp: float16 tensors required (CHECK_IS_HALF, ffmlp.cu:638-642)")

    @staticmethod
    def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                      forward_buffer, outputs):
        need_cuda(inputs, weights, forward_buffer, outputs); need_contig(inputs, weights, forward_buffer, outputs)
        _FFMLP._half(inputs, weights, forward_buffer, outputs)      # forward_buffer may be None: activations not saved
        check(_lib.load().lae_ffmlp_forward(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                            activation, output_activation, ptr(forward_buffer), ptr(outputs), stream()),
              "ffmlp_forward")

    @staticmethod
    def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                        inference_buffer, outputs):
        need_cuda(inputs, weights, outputs); need_contig(inputs, weights, outputs)
        _FFMLP._half(inputs, weights, outputs)
        check(_lib.load().lae_ffmlp_inference(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers,
                                              activation, output_activation, ptr(inference_buffer), ptr(outputs), stream()),
              "ffmlp_inference")

    @staticmethod
    def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                       output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, accumulate=False,
                       nonfinite_flag=None):
        """accumulate / nonfinite_flag (MI355X extensions, fused backward only): ADD the weight gradient to grad_weights (an
        optimizer-owned fp16 accumulator); address (int) of a device int32 OR-ed with 1 when a stored gradient is not finite"""
        ts = (grad, inputs, weights, forward_buffer, backward_buffer, grad_inputs, grad_weights)
        need_cuda(*ts); need_contig(*ts); _FFMLP._half(*ts)
        if accumulate or nonfinite_flag is not None:
            check(_lib.load().lae_ffmlp_backward_ex(ptr(grad), ptr(inputs), ptr(weights), ptr(forward_buffer), B, input_dim,
                                                    output_dim, hidden_dim, num_layers, activation, output_activation,
                                                    int(bool(calc_grad_inputs)), ptr(backward_buffer), ptr(grad_inputs),
                                                    ptr(grad_weights), int(bool(accumulate)), nonfinite_flag, stream()), "ffmlp_backward")
            return
        check(_lib.load().lae_ffmlp_backward(ptr(grad), ptr(inputs), ptr(weights), ptr(forward_buffer), B, input_dim,
                                             output_dim, hidden_dim, num_layers, activation, output_activation,
                                             int(bool(calc_grad_inputs)), ptr(backward_buffer), ptr(grad_inputs),
                                             ptr(grad_weights), stream()), "ffmlp_backward")

    @staticmethod
    def nerf_head_forward(enc, dirs, sigma_weights, color_weights, M, density_scale, h_out, sigmas, rgbs, level_major=False):
        """MI355X-native: network_ff.py:57-79 after the grid encoder, in one kernel (include/laenerf.h)"""
        ts = (enc, dirs, sigma_weights, color_weights, h_out, sigmas, rgbs)
        need_cuda(*ts); need_contig(*ts); _FFMLP._half(enc, sigma_weights, color_weights, h_out)
        check(_lib.load().lae_nerf_head_forward(ptr(enc), ptr(dirs), ptr(sigma_weights), ptr(color_weights), M,
                                                float(density_scale), ptr(h_out), ptr(sigmas), ptr(rgbs),
                                                int(bool(level_major)), stream()), "nerf_head_forward")

    @staticmethod
    def nerf_density_forward(enc, sigma_weights, M, density_scale, h_out, sigmas, level_major=False):
        need_cuda(enc, sigma_weights, h_out, sigmas); need_contig(enc, sigma_weights, h_out, sigmas)
        _FFMLP._half(enc, sigma_weights, h_out)
        check(_lib.load().lae_nerf_density_forward(ptr(enc), ptr(sigma_weights), M, float(density_scale), ptr(h_out),
                                                   ptr(sigmas), int(bool(level_major)), stream()), "nerf_density_forward")

    @staticmethod
    def nerf_head_backward(grad_sigmas, grad_rgbs, enc, dirs, h, rgbs, sigma_weights, color_weights, M, density_scale,
                           grad_h, grad_enc, grad_sigma_weights, grad_color_weights, accumulate=False, level_major=False,
                           nonfinite_flag=None):
        ts = (grad_sigmas, grad_rgbs, enc, dirs, h, rgbs, sigma_weights, color_weights, grad_h, grad_enc,
              grad_sigma_weights, grad_color_weights)
        need_cuda(*ts); need_contig(*ts)
        _FFMLP._half(enc, h, sigma_weights, color_weights, grad_h, grad_enc, grad_sigma_weights, grad_color_weights)
        # the deferred loss value of the criterion node whose gradient THIS call consumes rides in the reduction launch -- only on
        # the stream its partials were written on (another stream gives no ordering against them)
        key = _loss_key(grad_sigmas)
        pend = _pending_loss.get(key)
        if pend is not None and pend[5] != stream():
            pend = None
        if pend is not None:
            del _pending_loss[key]
        lp, ln, le, lsc, lo = pend[:5] if pend is not None else (None, 0, 0, None, None)
        try:
            check(_lib.load().lae_nerf_head_backward(ptr(grad_sigmas), ptr(grad_rgbs), ptr(enc), ptr(dirs), ptr(h), ptr(rgbs),
                                                     ptr(sigma_weights), ptr(color_weights), M, float(density_scale),
                                                     ptr(grad_h), ptr(grad_enc), ptr(grad_sigma_weights),
                                                     ptr(grad_color_weights), int(bool(accumulate)), int(bool(level_major)),
                                                     nonfinite_flag, ptr(lp), ln, le, ptr(lsc), ptr(lo), stream()), "nerf_head_backward")
        except RuntimeError:
            if pend is not None:
                _pending_loss[key] = pend             # the launch failed: the value is still unfinished, flush_pending_loss() can finish it
            raise
        if pend is not None:
            deferred_loss_stats["carried"] += 1

    @staticmethod
    def ffmlp_set_mode(mode):
        """0 = fused backward, wave-private dW with MFMA transposes (default); 1 = buffer-faithful three-kernel backward (fills
        forward/backward buffers); 3 = fused backward with the round-2 workgroup-cooperative dW kernel (include/laenerf.h)"""
        check(_lib.load().lae_ffmlp_set_mode(int(mode)), "ffmlp_set_mode")

    @staticmethod
    def fused_backward_available(input_dim, hidden_dim, num_layers, activation):
        return hidden_dim == 64 and activation == 0 and num_layers in (2, 3) and input_dim in (32, 48, 64)

    @staticmethod
    def allocate_splitk(size):
        check(_lib.load().lae_allocate_splitk(int(size)), "allocate_splitk")

    @staticmethod
    def free_splitk():
        check(_lib.load().lae_free_splitk(), "free_splitk")


# --------------------------------------------------------------------------- optional per-kernel timing
# HIP-event pairs recorded on torch's current stream (= the stream the kernels are launched on) around selected
# backend calls; used by bench.py for the roofline figure.  Off by default: zero overhead in the product path.
_timing = {"on": False, "only": None, "events": []}
_UNITS = {"grid_encode_forward": 4, "grid_encode_backward": 5, "grid_backward_plan": 2, "ffmlp_forward": 2, "ffmlp_inference": 2,
          "ffmlp_backward": 4, "nerf_head_forward": 4, "nerf_head_backward": 8, "nerf_density_forward": 2, "sh_encode_forward": 2, "march_rays_train": 6, "composite_rays_train_forward": 5,
          "composite_rays_train_backward": 9,
          "composite_rays_train_forward_blend": 5, "composite_rays_train_backward_blend": 9}


def enable_kernel_timing(on, only=("grid_encode_forward",)):
    """only = tuple of backend function names to time, or None for all of them.  While it is on, Python's collector is off (one
    explicit collection first): a full collection landing between an event pair puts a 160-184 us interval among 52 us ones, and
    where the allocation count trips the collector is an accident of the code that ran before (round 5: bench.py's `roofline`
    moved between 0.38 and 0.32 with unrelated edits)."""
    import gc
    if on and not _timing["on"]:
        _timing["gc_was_enabled"] = gc.isenabled()       # restored when the timing is switched off: a caller that runs
        gc.collect()                                     # with the collector off keeps it off
        gc.disable()
    elif not on and _timing["on"]:
        if _timing.pop("gc_was_enabled", True):
            gc.enable()
    _timing["on"], _timing["only"], _timing["events"] = bool(on), (set(only) if only else None), []


class kernel_timing:
    """`with backend.kernel_timing(only=...):` -- enable_kernel_timing(True, only) on entry, (False) on exit, also when the body
    raises (the collector state is restored either way); `.result` holds collect_kernel_timing() of the body"""

    def __init__(self, only=("grid_encode_forward",)):
        self.only, self.result = only, {}

    def __enter__(self):
        enable_kernel_timing(True, only=self.only)
        return self

    def __exit__(self, et, ev, tb):
        try:
            if et is None:
                self.result = collect_kernel_timing()
        finally:
            enable_kernel_timing(False)
        return False


def collect_kernel_timing():
    """-> {name: {ms (sum of event-pair durations), calls, units (sum of B / N / M of the calls)}}; synchronises"""
    torch.cuda.synchronize()
    out = {}
    each = {}
    for name, e0, e1, units in _timing["events"]:
        d = out.setdefault(name, {"ms": 0.0, "calls": 0, "units": 0})
        t = e0.elapsed_time(e1)
        d["ms"] += t; d["calls"] += 1; d["units"] += units
        each.setdefault(name, []).append(t)
    for name, ts in each.items():                          # spread of the intervals (a host stall inside one shows as max >> median)
        ts.sort()
        out[name]["median_ms"], out[name]["max_ms"], out[name]["min_ms"] = ts[len(ts) // 2], ts[-1], ts[0]
    _timing["events"] = []
    return out


def _wrap_timed(name, fn):
    ui = _UNITS.get(name)

    def timed(*a, **k):
        if not _timing["on"] or (_timing["only"] is not None and name not in _timing["only"]):
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # keep the stream busy while the host enqueues {e0, kernels, e1}: with an idle GPU the interval would also hold
        # the host's launch latency (Python + ctypes), which is not kernel time.  ~0.35 ms of spin (600 k shader cycles): rounds
        # 1-4 used 150 k (~85 us), which the encoder's wrapper -- the longest host path -- only just fitted under: unrelated
        # edits of bench.py moved its event time between 50 and 58 us while rocprofv3 kept seeing 49 us launches (round 5)
        torch.cuda._sleep(_timing.get("sleep_cycles", 600000))
        e0.record()
        r = fn(*a, **k)
        e1.record()
        _timing["events"].append((name, e0, e1, int(a[ui]) if ui is not None else 0))
        return r
    timed.__name__ = name
    return timed


class _Style:
    """LAENeRF palette recomposition (include/laenerf.h lae_palette_*; no reference extension: torch ops there)"""

    @staticmethod
    def palette_forward(w_logits, o_raw, palette, P, active_mask, M, pred, w_hat, o_hat):
        ts = (w_logits, o_raw, palette, pred, w_hat, o_hat)
        need_cuda(*ts); need_contig(*ts)
        if w_logits.dtype != _F16 or o_raw.dtype != _F16 or pred.dtype != _F16 or o_hat.dtype != _F16 or \
                palette.dtype != torch.float32 or w_hat.dtype != torch.float32:
            raise RuntimeError("palette_forward: MLP outputs / pred / o_hat float16, palette / w_hat float32")
        check(_lib.load().lae_palette_forward(ptr(w_logits), ptr(o_raw), ptr(palette), P, active_mask, M, ptr(pred), ptr(w_hat),
                                              ptr(o_hat), stream()), "palette_forward")

    @staticmethod
    def palette_backward(w_logits, o_raw, palette, P, active_mask, M, g_pred, g_w, g_o, g_w_logits, g_o_raw, g_palette):
        ts = (w_logits, o_raw, palette, g_pred, g_w, g_o, g_w_logits, g_o_raw, g_palette)
        need_cuda(*ts); need_contig(*ts)
        lib = _lib.load()
        ws = _workspace(w_logits.device, lib.lae_palette_backward_scratch_bytes(M))
        check(lib.lae_palette_backward(ptr(w_logits), ptr(o_raw), ptr(palette), P, active_mask, M, ptr(g_pred), ptr(g_w), ptr(g_o),
                                       ptr(g_w_logits), ptr(g_o_raw), ptr(g_palette), ptr(ws), stream()), "palette_backward")


    @staticmethod
    def style_assemble_forward(feats_lm, dirs, M, Mp, degree, feat, off_in, off_cols):
        """level-major encoder output + directions -> the two MLPs' input rows in one launch (include/laenerf.h)"""
        ts = (feats_lm, dirs, feat, off_in)
        need_cuda(*ts); need_contig(*ts)
        if feats_lm.dtype != _F16 or feat.dtype != _F16 or (off_in is not None and off_in.dtype != _F16) or (dirs is not None and dirs.dtype != torch.float32):
            raise RuntimeError("style_assemble_forward: features float16, directions float32")
        check(_lib.load().lae_style_assemble_forward(ptr(feats_lm), ptr(dirs), M, Mp, degree, ptr(feat), ptr(off_in), off_cols, stream()),
              "style_assemble_forward")

    @staticmethod
    def style_assemble_backward(g_feat, g_off, M, off_cols, grad_lm):
        ts = (g_feat, g_off, grad_lm)
        need_cuda(*ts); need_contig(*ts)
        if any(t is not None and t.dtype != _F16 for t in ts):
            raise RuntimeError("style_assemble_backward: float16 gradients")
        check(_lib.load().lae_style_assemble_backward(ptr(g_feat), ptr(g_off), M, off_cols, ptr(grad_lm), stream()), "style_assemble_backward")

    @staticmethod
    def style_loss_forward(pred, target, w_hat, o_hat, M, n_active, lw, scale, fin, reg_palette=None, reg_w=(0.0, 0.0)):
        """reg_palette [P,3] fp32 + reg_w = (palette_loss_valid, palette_loss_distinct): `palet_loss` joins the criterion (include/laenerf.h)"""
        ts = (pred, target, w_hat, o_hat, scale, fin, reg_palette)
        need_cuda(*ts); need_contig(*ts)
        assert fin.numel() >= 12
        lib = _lib.load()
        ws = _workspace(pred.device, lib.lae_style_loss_scratch_bytes(M))
        check(lib.lae_style_loss_forward(ptr(pred), ptr(target), ptr(w_hat), ptr(o_hat), M, n_active, float(lw[0]), float(lw[1]), float(lw[2]),
                                         ptr(scale), ptr(fin), ptr(ws), ptr(reg_palette), 0 if reg_palette is None else reg_palette.shape[0],
                                         float(reg_w[0]), float(reg_w[1]), stream()), "style_loss_forward")

    @staticmethod
    def style_loss_backward(w_logits, o_raw, palette, P, active_mask, M, target, fin, upstream, lw, g_w_logits, g_o_raw, g_palette,
                            reg_w=None, accumulate=False):
        """accumulate: g_palette += (LAE_STYLE_ACCUMULATE_PALETTE) -- the caller's persistent fp32 gradient buffer"""
        ts = (w_logits, o_raw, palette, target, fin, upstream, g_w_logits, g_o_raw, g_palette)
        need_cuda(*ts); need_contig(*ts)
        lib = _lib.load()
        ws = _workspace(w_logits.device, max(lib.lae_palette_backward_scratch_bytes(M), lib.lae_style_loss_scratch_bytes(M)))
        check(lib.lae_style_loss_backward(ptr(w_logits), ptr(o_raw), ptr(palette), P, active_mask, M, ptr(target), ptr(fin), ptr(upstream),
                                          float(lw[0]), float(lw[1]), float(lw[2]), ptr(g_w_logits), ptr(g_o_raw), ptr(g_palette), ptr(ws),
                                          int(reg_w is not None) | (2 if accumulate else 0), float(reg_w[0]) if reg_w else 0.0, float(reg_w[1]) if reg_w else 0.0,
                                          stream()), "style_loss_backward")


style_backend = _Style

for _cls in (_RayMarching, _GridEncoder, _SHEncoder, _FFMLP):
    for _k, _v in list(vars(_cls).items()):
        if isinstance(_v, staticmethod) and not _k.startswith("_") and _k not in ("fused_backward_available", "ffmlp_set_mode", "set_backward_mode",
                                                                                   "allocate_splitk", "free_splitk", "render_frame_last_status",
                                                                                   "render_frame_check"):
            setattr(_cls, _k, staticmethod(_wrap_timed(_k, _v.__func__)))

raymarching_backend = _RayMarching
gridencoder_backend = _GridEncoder
shencoder_backend = _SHEncoder
ffmlp_backend = _FFMLP
freqencoder_backend = _FreqEncoder


def _as_module(name, cls):
    m = types.ModuleType(name)
    for k, v in vars(cls).items():
        if not k.startswith("_") and isinstance(v, staticmethod):
            setattr(m, k, v.__func__)
    m.__doc__ = f"laenerf_amd HIP backend standing in for the reference's `{name}` extension"
    return m


def install_as_reference_backends():
    """Make `import _raymarching / _gridencoder / _shencoder / _ffmlp` resolve to the HIP backend, so the
    reference's own Python (raymarching.py, grid.py, sphere_harmonics.py, ffmlp.py, nerf/renderer.py,
    editing/*) runs on it unmodified."""
    _lib.load()
    for name, cls in (("_raymarching", _RayMarching), ("_gridencoder", _GridEncoder), ("_shencoder", _SHEncoder),
                      ("_ffmlp", _FFMLP), ("_freqencoder", _FreqEncoder)):
        sys.modules[name] = _as_module(name, cls)
