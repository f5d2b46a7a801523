"""CPU backend modules with the reference's pybind11 signatures, implemented on the oracle.

TEST INFRASTRUCTURE (used only by tests/golden/make_golden.py, in the build container).  They let the
reference's UNMODIFIED Python (raymarching.py, grid.py, sphere_harmonics.py, ffmlp.py, nerf/renderer.py,
nerf/network_ff.py) run on CPU tensors: `sys.modules['_raymarching'] = ...` is what those wrappers import
first (raymarching/raymarching.py:9-12).  Tensors are passed to liblae_oracle.so by raw pointer, so the
in-place semantics of the backend calls are preserved.
"""
import ctypes
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle as O  # noqa: E402

u32, f32c, i32c, vp = ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_void_p


def _p(t):
    if t is None:
        return None
    assert not t.is_cuda and t.is_contiguous()
    return vp(t.data_ptr())


def make_raymarching():
    m = types.ModuleType("_raymarching")
    L = O.lib()

    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        L.orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32c(min_near), _p(nears), _p(fars))

    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        L.orc_sph_from_ray(_p(rays_o), _p(rays_d), f32c(radius), u32(N), _p(coords))

    def morton3D(coords, N, indices):
        L.orc_morton3D(_p(coords.contiguous()), u32(N), _p(indices))

    def morton3D_invert(indices, N, coords):
        L.orc_morton3D_invert(_p(indices.contiguous()), u32(N), _p(coords))

    def packbits(grid, N, thresh, bitfield):
        L.orc_packbits(_p(grid), u32(N), f32c(thresh), _p(bitfield))

    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                         rays, counter, noises):
        L.orc_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), f32c(bound), f32c(dt_gamma), u32(max_steps), u32(N),
                               u32(C), u32(H), u32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays),
                               _p(counter), _p(noises))

    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        L.orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N), f32c(T_thresh),
                                           _p(weights_sum), _p(depth), _p(image))

    def composite_rays_train_backward(gws, gimg, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, gs, gc):
        L.orc_composite_rays_train_backward(_p(gws), _p(gimg), _p(sigmas), _p(rgbs), _p(deltas), _p(rays),
                                            _p(weights_sum), _p(image), u32(M), u32(N), f32c(T_thresh), _p(gs), _p(gc))

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears,
                   fars, xyzs, dirs, deltas, noises):
        L.orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), f32c(bound),
                         f32c(dt_gamma), u32(max_steps), u32(C), u32(H), _p(grid), None, _p(nears), _p(fars), _p(xyzs),
                         _p(dirs), _p(deltas), None, _p(noises))

    def march_rays_distill(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid,
                           edit_grid, nears, fars, xyzs, dirs, deltas, int_edit, noises):
        L.orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), f32c(bound),
                         f32c(dt_gamma), u32(max_steps), u32(C), u32(H), _p(grid), _p(edit_grid), _p(nears), _p(fars),
                         _p(xyzs), _p(dirs), _p(deltas), _p(int_edit), _p(noises))

    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights, depth, image):
        L.orc_composite_rays(u32(n_alive), u32(n_step), f32c(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas.contiguous()),
                             _p(rgbs.contiguous()), _p(deltas), _p(weights), None, _p(depth), None, None, _p(image))

    def composite_rays_distill(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights, weights_edit,
                               depth, depth_edit, int_edit, image):
        L.orc_composite_rays(u32(n_alive), u32(n_step), f32c(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas.contiguous()),
                             _p(rgbs.contiguous()), _p(deltas), _p(weights), _p(weights_edit), _p(depth), _p(depth_edit),
                             _p(int_edit), _p(image))

    for k, v in list(locals().items()):
        if callable(v) and not k.startswith("_") and k not in ("m", "L"):
            setattr(m, k, v)
    return m


def make_gridencoder():
    m = types.ModuleType("_gridencoder")
    L = O.lib()

    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L_, S, H, dy_dx, gridtype, align_corners, interp):
        f16 = embeddings.dtype == torch.float16
        L.orc_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), u32(B), u32(D), u32(C), u32(L_),
                                  f32c(S), u32(H), _p(dy_dx), u32(gridtype), i32c(int(align_corners)), u32(interp),
                                  i32c(int(f16)), i32c(0))

    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L_, S, H, dy_dx, grad_inputs,
                             gridtype, align_corners, interp):
        f16 = grad.dtype == torch.float16
        L.orc_grid_encode_backward(_p(grad), _p(inputs), _p(offsets), _p(grad_embeddings), u32(B), u32(D), u32(C), u32(L_),
                                   f32c(S), u32(H), _p(dy_dx), _p(grad_inputs), u32(gridtype), i32c(int(align_corners)),
                                   u32(interp), i32c(int(f16)), i32c(0))

    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L_, S, H, gridtype, align_corners):
        L.orc_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), f32c(weight), u32(B), u32(D), u32(C),
                                   u32(L_), f32c(S), u32(H), u32(gridtype), i32c(int(align_corners)))

    m.grid_encode_forward, m.grid_encode_backward, m.grad_total_variation = grid_encode_forward, grid_encode_backward, grad_total_variation
    return m


def make_shencoder():
    m = types.ModuleType("_shencoder")
    L = O.lib()

    def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
        L.orc_sh_encode_forward(_p(inputs), _p(outputs), u32(B), u32(D), u32(C), _p(dy_dx))

    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        L.orc_sh_encode_backward(_p(grad), u32(B), u32(D), u32(C), _p(dy_dx), _p(grad_inputs))

    m.sh_encode_forward, m.sh_encode_backward = sh_encode_forward, sh_encode_backward
    return m


def make_ffmlp():
    m = types.ModuleType("_ffmlp")
    L = O.lib()

    def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                      forward_buffer, outputs):
        L.orc_ffmlp_forward(_p(inputs), _p(weights), u32(B), u32(input_dim), u32(output_dim), u32(hidden_dim),
                            u32(num_layers), u32(activation), u32(output_activation), _p(forward_buffer), _p(outputs))

    def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                        inference_buffer, outputs):
        L.orc_ffmlp_forward(_p(inputs), _p(weights), u32(B), u32(input_dim), u32(output_dim), u32(hidden_dim),
                            u32(num_layers), u32(activation), u32(output_activation), None, _p(outputs))

    def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                       output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights):
        L.orc_ffmlp_backward(_p(grad), _p(inputs), _p(weights), _p(forward_buffer), u32(B), u32(input_dim), u32(output_dim),
                             u32(hidden_dim), u32(num_layers), u32(activation), i32c(int(calc_grad_inputs)),
                             _p(backward_buffer), _p(grad_inputs) if calc_grad_inputs else None, _p(grad_weights))

    m.ffmlp_forward, m.ffmlp_inference, m.ffmlp_backward = ffmlp_forward, ffmlp_inference, ffmlp_backward
    m.allocate_splitk = lambda size: None
    m.free_splitk = lambda: None
    return m


def install():
    sys.modules["_raymarching"] = make_raymarching()
    sys.modules["_gridencoder"] = make_gridencoder()
    sys.modules["_shencoder"] = make_shencoder()
    sys.modules["_ffmlp"] = make_ffmlp()
