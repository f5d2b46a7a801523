// densitygrid.hip -- occupancy-grid maintenance for gfx950.
//
// The reference keeps this part in Python (nerf/renderer.py:482-649: mark_untrained_grid, update_extra_state): per
// cascade ~40 torch ops (arange/meshgrid/cat/morton/scale/jitter/.../index_put/mask/maximum) around the density
// query.  Here the producers and consumers of the density query are three small kernels:
//   k_positions      cell -> jittered world position + Morton index          (renderer.py:580-592, 602-621)
//   k_scatter_max    sigma -> per-cell maximum in an integer scratch grid    (renderer.py:596, 627: tmp_grid[idx] = sigma)
//   k_ema            density_grid = max(density_grid * decay, tmp) on cells that were sampled and are trainable
//                                                                             (renderer.py:633-634)
// and the camera-coverage test of mark_untrained_grid is one kernel with the poses in LDS (renderer.py:482-554).
// Compiled with -ffp-contract=off; every expression follows the reference's fp32 evaluation order.
#include "lae_common.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {   // raymarching.cu:59-66
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

// cell centre in [-1,1]:  2 * c / (H - 1) - 1   (renderer.py:582)
__device__ __forceinline__ float cell_unit(int32_t c, float hm1) { return (2.0f * (float)c) / hm1 - 1.0f; }

// One thread per point.  coords == NULL: point j is cell (j / H^2, (j / H) % H, j % H) -- the meshgrid order of the
// full sweep; otherwise coords [n,3] int32 (the random + occupied cells of the partial sweep).
// xyz = unit * (bound_c - hgs) + (noise * 2 - 1) * hgs   (renderer.py:585-590); noise == NULL -> no jitter.
__global__ void k_positions(const int32_t* __restrict__ coords, uint32_t n, uint32_t H, float scale, float hgs,
                            const float* __restrict__ noise, float* __restrict__ xyzs, int32_t* __restrict__ indices) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int32_t c[3];
    if (coords) { c[0] = coords[3 * (size_t)j]; c[1] = coords[3 * (size_t)j + 1]; c[2] = coords[3 * (size_t)j + 2]; }
    else { c[0] = (int32_t)(j / (H * H)); c[1] = (int32_t)((j / H) % H); c[2] = (int32_t)(j % H); }
    const float hm1 = (float)(H - 1);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float p = cell_unit(c[k], hm1) * scale;
        if (noise) p = p + (noise[3 * (size_t)j + k] * 2.0f - 1.0f) * hgs;
        xyzs[3 * (size_t)j + k] = p;
    }
    indices[j] = (int32_t)morton_encode((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]);
}

// tmp holds (bits(sigma) + 1) of the largest sigma that landed in the cell, 0 = not sampled.  Non-negative floats
// order like their bit patterns, so atomicMax on uint32 is an exact float max.  Where the reference's
// `tmp_grid[cas, indices] = sigmas` meets duplicate indices (partial sweep) its winner is arbitrary; max is one of
// the values it can produce and is deterministic.
__global__ void k_scatter_max(const float* __restrict__ sigmas, const int32_t* __restrict__ indices, uint32_t n,
                              float density_scale, uint32_t cells, uint32_t* __restrict__ tmp) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float v = sigmas[j] * density_scale;
    const uint32_t idx = (uint32_t)indices[j];
    if (!(v >= 0.0f) || idx >= cells) return;                    // tmp_grid >= 0 test of renderer.py:633 (NaN fails too)
    atomicMax(&tmp[idx], __float_as_uint(v + 0.0f) + 1u);
}

__global__ void k_ema(float* __restrict__ grid, uint32_t* __restrict__ tmp, uint32_t cells, float decay) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const uint32_t t = tmp[i];
    if (t == 0) return;
    tmp[i] = 0;                                                  // scratch is left zeroed for the next update
    const float g = grid[i];
    if (g >= 0.0f) grid[i] = fmaxf(g * decay, __uint_as_float(t - 1u));
}

// mark_untrained_grid: one thread per (cascade, cell); poses [B,4,4] row-major camera-to-world.
constexpr int POSE_CHUNK = 256;
__global__ __launch_bounds__(256) void k_mark_untrained(const float* __restrict__ poses, uint32_t B, float kx, float ky,
                                                        uint32_t C, uint32_t H, float bound, float min_near,
                                                        int filter_close, float* __restrict__ grid) {
    __shared__ float P[POSE_CHUNK * 12];
    const uint32_t cells = H * H * H;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cas = blockIdx.y;
    const bool live = gid < cells;
    const uint32_t j = live ? gid : 0;
    const int32_t c[3] = {(int32_t)(j / (H * H)), (int32_t)((j / H) % H), (int32_t)(j % H)};
    const float bound_c = fminf((float)(1u << cas), bound);
    const float hgs = bound_c / (float)H;
    const float hm1 = (float)(H - 1);
    float w[3];
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = cell_unit(c[k], hm1) * (bound_c - hgs);
    const float margin = hgs * 2.0f;
    uint32_t count = 0, close = 0;
    for (uint32_t b0 = 0; b0 < B; b0 += POSE_CHUNK) {
        const uint32_t nb = min((uint32_t)POSE_CHUNK, B - b0);
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < nb * 12; e += blockDim.x) P[e] = poses[(size_t)(b0 + e / 12) * 16 + (e % 12)];
        __syncthreads();
        for (uint32_t b = 0; b < nb; b++) {
            const float* Q = P + b * 12;                         // rows 0..2 of [R | t]
            const float dx = w[0] - Q[3], dy = w[1] - Q[7], dz = w[2] - Q[11];
            // cam = (world - t) @ R   (renderer.py:530-531)
            const float cx = dx * Q[0] + dy * Q[4] + dz * Q[8];
            const float cy = dx * Q[1] + dy * Q[5] + dz * Q[9];
            const float cz = dx * Q[2] + dy * Q[6] + dz * Q[10];
            const bool in = (cz > 0.0f) && (fabsf(cx) < kx * cz + margin) && (fabsf(cy) < ky * cz + margin);
            count += in;
            close += in && (cz < min_near);
            if (filter_close) close += sqrtf(cx * cx + cy * cy + cz * cz) < min_near;
        }
    }
    if (live && (count == 0 || close != 0)) grid[(size_t)cas * cells + morton_encode(c[0], c[1], c[2])] = -1.0f;
}

}  // namespace

extern "C" {

int lae_density_grid_positions(const int32_t* coords, uint32_t n, uint32_t H, float bound_c, const float* noise,
                               float* xyzs, int32_t* indices, void* stream) {
    if (n == 0) return LAE_OK;
    if (!xyzs || !indices) return LAE_ENULL;
    if (H < 2 || H > 1024 || (!coords && (uint64_t)n > (uint64_t)H * H * H)) return LAE_EINVAL;
    const float hgs = bound_c / (float)H;
    k_positions<<<lae::cdiv(n, 256), 256, 0, STREAM(stream)>>>(coords, n, H, bound_c - hgs, hgs, noise, xyzs, indices);
    return lae::check_launch("density_grid_positions");
}

int lae_density_grid_update(const float* sigmas, const int32_t* indices, uint32_t n, float density_scale, float decay,
                            uint32_t cells, float* grid, uint32_t* tmp, void* stream) {
    if (cells == 0) return LAE_OK;
    if (!grid || !tmp || (n && (!sigmas || !indices))) return LAE_ENULL;
    if (n) k_scatter_max<<<lae::cdiv(n, 256), 256, 0, STREAM(stream)>>>(sigmas, indices, n, density_scale, cells, tmp);
    k_ema<<<lae::cdiv(cells, 256), 256, 0, STREAM(stream)>>>(grid, tmp, cells, decay);
    return lae::check_launch("density_grid_update");
}

int lae_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t C, uint32_t H,
                            float bound, float min_near, int filter_close_point, float* grid, void* stream) {
    if (C == 0 || H == 0) return LAE_OK;
    if (!grid || (B && !poses)) return LAE_ENULL;
    if (H < 2 || H > 1024 || C > 16) return LAE_EINVAL;
    // cx / fx and cy / fy are formed in double by the Python caller and then used as fp32 scalars
    const float kx = (float)((double)cx / (double)fx), ky = (float)((double)cy / (double)fy);
    dim3 grid_dim(lae::cdiv((uint64_t)H * H * H, 256), C);
    k_mark_untrained<<<grid_dim, 256, 0, STREAM(stream)>>>(poses, B, kx, ky, C, H, bound, min_near, filter_close_point, grid);
    return lae::check_launch("mark_untrained_grid");
}

}  // extern "C"
