"""numpy front-end of the CPU oracle (oracle/lae_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the laenerf_amd package.

Every function takes / returns numpy arrays and mirrors one backend function of
the reference (`_raymarching`, `_gridencoder`, `_shencoder`, `_ffmlp`); outputs
are allocated here the way the reference's autograd.Functions allocate them
(raymarching/raymarching.py, gridencoder/grid.py, ...).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblae_oracle.so")
_SO_NOFMA = os.path.join(_HERE, "liblae_oracle_nofma.so")


def build(force=False):
    src = os.path.join(_HERE, "lae_oracle.c")
    stale = any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in (_SO, _SO_NOFMA))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_libs = {}
_flavour = "fma"


def lib():
    """the oracle library of the active flavour ("fma": FMA(a,b,c) = fmaf, the default and the one every parity test
    uses; "nofma": a*b+c with two roundings -- only for the contraction-independence tests)."""
    if _flavour not in _libs:
        build()
        so = ctypes.CDLL(_SO if _flavour == "fma" else _SO_NOFMA)
        so.orc_flavour_fma.restype = ctypes.c_int
        assert so.orc_flavour_fma() == (1 if _flavour == "fma" else 0)
        _libs[_flavour] = so
    return _libs[_flavour]


class flavour:
    """`with oracle.flavour("nofma"): ...` runs the enclosed oracle calls on the un-fused build."""

    def __init__(self, name):
        assert name in ("fma", "nofma"), name
        self.name = name

    def __enter__(self):
        global _flavour
        self.prev, _flavour = _flavour, self.name
        return self

    def __exit__(self, *exc):
        global _flavour
        _flavour = self.prev
        return False


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


u32, f32c, i32c = ctypes.c_uint32, ctypes.c_float, ctypes.c_int


# ------------------------------------------------------------------ fp16 helpers
def to_f16_bits(x):
    x = _f32(x)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().orc_f32_to_f16(_p(x), _p(out), ctypes.c_uint64(x.size))
    return out


def from_f16_bits(h):
    h = np.ascontiguousarray(h, dtype=np.uint16)
    out = np.empty(h.shape, dtype=np.float32)
    lib().orc_f16_to_f32(_p(h), _p(out), ctypes.c_uint64(h.size))
    return out


# ------------------------------------------------------------------ raymarching
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32c(min_near), _p(nears), _p(fars))
    return nears, fars


def get_rays(poses, intrinsics, H, W, inds=None, offset=None):
    """nerf/utils.py:61-153 for given pixel indices: poses [B,4,4], inds None (all pixels) | [N] | [B,N] int64,
    offset None | (off_x, off_y) (perturb_ray_dirs) -> rays_o, rays_d [B,N,3]"""
    poses = _f32(poses).reshape(-1, 4, 4)
    B = poses.shape[0]
    fx, fy, cx, cy = (float(np.float32(v)) for v in intrinsics)
    if inds is None:
        N, ip, stride = H * W, None, 0
    else:
        inds = np.ascontiguousarray(inds, dtype=np.int64)
        N = inds.shape[-1]
        stride = N if inds.ndim == 2 and inds.shape[0] == B and B > 1 else 0
        ip = _p(inds)
    rays_o, rays_d = np.empty((B, N, 3), np.float32), np.empty((B, N, 3), np.float32)
    ox, oy = (0.0, 0.0) if offset is None else (float(offset[0]), float(offset[1]))
    lib().orc_get_rays(_p(poses), u32(B), f32c(fx), f32c(fy), f32c(cx), f32c(cy), u32(H), u32(W), ip, ctypes.c_uint64(stride),
                       u32(N), ctypes.c_int(0 if offset is None else 1), f32c(ox), f32c(oy), _p(rays_o), _p(rays_d))
    return rays_o, rays_d


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), f32c(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords)
    N = coords.shape[0]
    out = np.empty(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices)
    N = indices.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32c(thresh), _p(out))
    return out


def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, noises, M=None,
                     dt_gamma=0.0, max_steps=1024, counter=None):
    """Backend-level call: returns (xyzs[M,3], dirs[M,3], deltas[M,2], rays[N,3], counter[2])."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    if M is None:
        M = N * max_steps
    bitfield, nears, fars, noises = _u8(bitfield), _f32(nears), _f32(fars), _f32(noises)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32) if counter is None else _i32(counter)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(bitfield), f32c(bound), f32c(dt_gamma), u32(max_steps),
                               u32(N), u32(C), u32(H), u32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                               _p(deltas), _p(rays), _p(counter), _p(noises))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N),
                                           f32c(T_thresh), _p(ws), _p(depth), _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, ws, image, T_thresh=1e-4):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    grad_ws, grad_image, ws, image = _f32(grad_ws), _f32(grad_image), _f32(ws), _f32(image)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(grad_ws), _p(grad_image), _p(sigmas), _p(rgbs), _p(deltas),
                                            _p(rays), _p(ws), _p(image), u32(M), u32(N), f32c(T_thresh),
                                            _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars,
               noises, align=-1, dt_gamma=0.0, max_steps=1024, edit_bitfield=None):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    edit_occ = np.zeros(M, np.uint8) if edit_bitfield is not None else None
    rays_alive, rays_t, bitfield = _i32(rays_alive), _f32(rays_t), _u8(bitfield)
    eb = _u8(edit_bitfield) if edit_bitfield is not None else None
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d),
                         f32c(bound), f32c(dt_gamma), u32(max_steps), u32(C), u32(H), _p(bitfield), _p(eb),
                         _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(edit_occ), _p(noises))
    if edit_bitfield is not None:
        return xyzs, dirs, deltas, edit_occ
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                   T_thresh=1e-2, weights_edit_sum=None, depth_edit=None, edit_occ=None):
    """In place on rays_alive, rays_t, weights_sum, depth, image (+ edit accumulators): arrays must be
    C-contiguous with the right dtype (they are mutated)."""
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32),
                  (depth, np.float32), (image, np.float32)):
        assert a.dtype == dt and a.flags.c_contiguous
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    eo = _u8(edit_occ) if edit_occ is not None else None
    lib().orc_composite_rays(u32(n_alive), u32(n_step), f32c(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas),
                             _p(rgbs), _p(deltas), _p(weights_sum), _p(weights_edit_sum), _p(depth),
                             _p(depth_edit), _p(eo), _p(image))


# ------------------------------------------------------------------ gridencoder
def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, align_corners=False):
    """Level sizing of GridEncoder.__init__ (gridencoder/grid.py:98-127)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_dy_dx=False,
                        gridtype=0, align_corners=False, interp=0, f16=False, out_blc=False):
    """embeddings: float32 array (f16=False) or uint16 fp16 bits (f16=True).  Returns outputs in the
    backend layout [L,B,C] (or [B,L*C] with out_blc) and dy_dx [B, L*D*C] or None."""
    inputs = _f32(inputs)
    B, D = inputs.shape
    offsets = _i32(offsets)
    L = offsets.shape[0] - 1
    C = embeddings.shape[1]
    S = np.float32(np.log2(per_level_scale))
    dt = np.uint16 if f16 else np.float32
    emb = np.ascontiguousarray(embeddings, dtype=dt)
    out = np.empty((B, L * C) if out_blc else (L, B, C), dt)
    dy_dx = np.empty((B, L * D * C), dt) if calc_dy_dx else None
    lib().orc_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(out), u32(B), u32(D), u32(C), u32(L),
                                  f32c(S), u32(base_resolution), _p(dy_dx), u32(gridtype), i32c(int(align_corners)),
                                  u32(interp), i32c(int(f16)), i32c(int(out_blc)))
    return out, dy_dx


def grid_encode_backward(grad, inputs, embeddings_shape, offsets, per_level_scale, base_resolution, dy_dx=None,
                         gridtype=0, align_corners=False, interp=0, f16=False, grad_blc=False):
    inputs = _f32(inputs)
    B, D = inputs.shape
    offsets = _i32(offsets)
    L = offsets.shape[0] - 1
    C = embeddings_shape[1]
    S = np.float32(np.log2(per_level_scale))
    dt = np.uint16 if f16 else np.float32
    grad = np.ascontiguousarray(grad, dtype=dt)
    g_emb = np.zeros(embeddings_shape, dt)
    g_in = np.zeros((B, D), dt) if dy_dx is not None else None
    dd = np.ascontiguousarray(dy_dx, dtype=dt) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(grad), _p(inputs), _p(offsets), _p(g_emb), u32(B), u32(D), u32(C), u32(L),
                                   f32c(S), u32(base_resolution), _p(dd), _p(g_in), u32(gridtype),
                                   i32c(int(align_corners)), u32(interp), i32c(int(f16)), i32c(int(grad_blc)))
    return g_emb, g_in


def grad_total_variation(inputs, embeddings, grad, offsets, weight, per_level_scale, base_resolution,
                         gridtype=0, align_corners=False):
    inputs, embeddings = _f32(inputs), _f32(embeddings)
    assert grad.dtype == np.float32 and grad.flags.c_contiguous
    B, D = inputs.shape
    offsets = _i32(offsets)
    L = offsets.shape[0] - 1
    C = embeddings.shape[1]
    S = np.float32(np.log2(per_level_scale))
    lib().orc_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), f32c(weight), u32(B), u32(D),
                                   u32(C), u32(L), f32c(S), u32(base_resolution), u32(gridtype),
                                   i32c(int(align_corners)))


# ------------------------------------------------------------------ shencoder
def sh_encode_forward(inputs, degree, calc_dy_dx=False):
    inputs = _f32(inputs)
    B, D = inputs.shape
    out = np.empty((B, degree * degree), np.float32)
    dy_dx = np.empty((B, D * degree * degree), np.float32) if calc_dy_dx else None
    lib().orc_sh_encode_forward(_p(inputs), _p(out), u32(B), u32(D), u32(degree), _p(dy_dx))
    return out, dy_dx


def sh_encode_backward(grad, dy_dx, degree, D=3):
    grad, dy_dx = _f32(grad), _f32(dy_dx)
    B = grad.shape[0]
    g_in = np.zeros((B, D), np.float32)
    lib().orc_sh_encode_backward(_p(grad), u32(B), u32(D), u32(degree), _p(dy_dx), _p(g_in))
    return g_in


# ------------------------------------------------------------------ freqencoder
def freq_encode_forward(inputs, degree):
    """freqencoder/freq.py:15-33: [B, D] -> [B, D + 2*D*degree]"""
    inputs = _f32(inputs)
    B, D = inputs.shape
    C = D + 2 * D * degree
    out = np.empty((B, C), np.float32)
    lib().orc_freq_encode_forward(_p(inputs), u32(B), u32(D), u32(degree), u32(C), _p(out))
    return out


def freq_encode_backward(grad, outputs, D, degree):
    grad, outputs = _f32(grad), _f32(outputs)
    B, C = grad.shape
    g_in = np.zeros((B, D), np.float32)
    lib().orc_freq_encode_backward(_p(grad), _p(outputs), u32(B), u32(D), u32(degree), u32(C), _p(g_in))
    return g_in


# ------------------------------------------------------------------ ffmlp
def ffmlp_num_params(input_dim, hidden_dim, num_layers, padded_output_dim=16):
    return hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + padded_output_dim)


def ffmlp_forward(inputs_h, weights_h, input_dim, output_dim, hidden_dim, num_layers, activation=0,
                  output_activation=6, want_buffer=True):
    """inputs_h [B,in] / weights_h flat: uint16 fp16 bits.  Returns (outputs[B,out] u16, forward_buffer u16|None)."""
    inputs_h = np.ascontiguousarray(inputs_h, np.uint16)
    weights_h = np.ascontiguousarray(weights_h, np.uint16)
    B = inputs_h.shape[0]
    out = np.empty((B, output_dim), np.uint16)
    fb = np.empty((num_layers, B, hidden_dim), np.uint16) if want_buffer else None
    lib().orc_ffmlp_forward(_p(inputs_h), _p(weights_h), u32(B), u32(input_dim), u32(output_dim), u32(hidden_dim),
                            u32(num_layers), u32(activation), u32(output_activation), _p(fb), _p(out))
    return out, fb


def ffmlp_backward(grad_h, inputs_h, weights_h, forward_buffer, input_dim, output_dim, hidden_dim, num_layers,
                   activation=0, calc_grad_inputs=False):
    grad_h = np.ascontiguousarray(grad_h, np.uint16)
    inputs_h = np.ascontiguousarray(inputs_h, np.uint16)
    weights_h = np.ascontiguousarray(weights_h, np.uint16)
    forward_buffer = np.ascontiguousarray(forward_buffer, np.uint16)
    B = inputs_h.shape[0]
    bb = np.zeros((num_layers, B, hidden_dim), np.uint16)
    gi = np.zeros((B, input_dim), np.uint16) if calc_grad_inputs else None
    gw = np.zeros(weights_h.shape, np.uint16)
    lib().orc_ffmlp_backward(_p(grad_h), _p(inputs_h), _p(weights_h), _p(forward_buffer), u32(B), u32(input_dim),
                             u32(output_dim), u32(hidden_dim), u32(num_layers), u32(activation),
                             i32c(int(calc_grad_inputs)), _p(bb), _p(gi), _p(gw))
    return gw, gi, bb


# ---------------------------------------------------------------- occupancy-grid maintenance (renderer.py:482-649)
def density_grid_positions(n, H, bound_c, noise=None, coords=None):
    xyzs = np.empty((n, 3), np.float32)
    idx = np.empty(n, np.int32)
    coords = None if coords is None else _i32(coords)
    noise = None if noise is None else _f32(noise)
    lib().orc_density_grid_positions(None if coords is None else _p(coords), u32(n), u32(H), f32c(bound_c),
                                     None if noise is None else _p(noise), _p(xyzs), _p(idx))
    return xyzs, idx


def density_grid_update(grid_c, sigmas, indices, density_scale=1.0, decay=0.95, rule=0):
    """-> updated copy of one cascade's grid; rule 0 = last write wins (torch CPU), 1 = maximum wins (HIP kernel)"""
    g = _f32(grid_c).copy()
    sigmas, indices = _f32(sigmas), _i32(indices)
    lib().orc_density_grid_update(_p(sigmas), _p(indices), u32(sigmas.size), f32c(density_scale), f32c(decay), u32(g.size),
                                  _p(g), ctypes.c_int(rule))
    return g


def mark_untrained_grid(grid, poses, intrinsics, bound, min_near=0.2, filter_close_point=False, H=128):
    """grid [C, H^3] -> (marked copy, per-cell distance of the closest test from its decision boundary)"""
    g = _f32(grid).copy()
    poses = _f32(poses)
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    margin = np.empty(g.shape, np.float32)
    lib().orc_mark_untrained_grid(_p(poses), u32(poses.shape[0]), f32c(fx), f32c(fy), f32c(cx), f32c(cy), u32(g.shape[0]),
                                  u32(H), f32c(bound), f32c(min_near), ctypes.c_int(int(filter_close_point)), _p(g), _p(margin))
    return g, margin


# ------------------------------------------------------------------ edit-grid region growing (editing/editgrid.py)
def grow_region_queue(grid, density_grid, density_thresh, queue, grow_iterations=5000, H=128, max_n=32):
    """EditGrid.grow_region_queue (editing/editgrid.py:274-340) restated on numpy arrays.

    grid: uint8 [C*H^3/8] selection bitfield (a modified copy is returned); density_grid: float32 [C, H^3] (Morton order);
    queue: sequence of (x, y, z, level).  Returns (grid, remaining queue as a list, cells popped).
    The indexed byte assignment of set_edit_bitfield_at (:35-38) has CPU-tensor semantics here: every right-hand side
    is read before any write and, for duplicate byte indices inside one batch, the last element wins."""
    from collections import deque
    grid = np.array(grid, dtype=np.uint8, copy=True)
    density_grid = _f32(density_grid)
    V = H ** 3
    q = deque((int(x), int(y), int(z), int(l)) for x, y, z, l in queue)
    offs = ((-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0))                 # :314-321
    ctr = 0
    while ctr < grow_iterations and q:
        num = min(max_n, len(q), grow_iterations - ctr)                                         # :289
        batch = [q.popleft() for _ in range(num)]
        coords = np.array([b[:3] for b in batch], np.int32)
        lvl = np.array([b[3] for b in batch], np.int64)
        pos = morton3D(coords).astype(np.int64) % V                                             # :299-301
        byte = pos // 8 + (V * lvl) // 8
        bit = pos % 8
        dens = density_grid[lvl, pos]
        old = grid[byte]
        cond = (dens >= density_thresh) & (((old >> bit) & 1) == 0)                             # :303-308
        if cond.any():
            new = (old & ~(1 << bit).astype(np.uint8)) | (1 << bit).astype(np.uint8)            # :35-38, value = 1
            for i in np.nonzero(cond)[0]:                                                       # in order: the last write to a byte stays
                grid[byte[i]] = new[i]
            for i in np.nonzero(cond)[0]:
                for o in offs:
                    c = coords[i] + np.array(o, np.int32)
                    if (c >= 0).all() and (c < H).all():                                        # :327-329
                        q.append((int(c[0]), int(c[1]), int(c[2]), int(lvl[0])))                # :323 level of the batch's first cell
        ctr += num
    return grid, list(q), ctr
