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


// ---- partial sweep of update_extra_state on the device (renderer.py:600-612).  The reference draws n random cells and n of
// the OCCUPIED cells: `occ = nonzero(density_grid[cas] > 0); occ[randint(len(occ), n)]` -- a boolean-mask compaction whose length
// the host reads back (a device sync per cascade), a host-sized randint, an index, a morton3D_invert and a cat.  Here:
//   k_occ_count    occupied cells per 1024-cell chunk
//   k_occ_scan     exclusive prefix over the <= 2^20 chunk counts (one workgroup), total K
//   k_occ_compact  occ_list[prefix + rank in chunk] = cell, in index order (= nonzero's order)
//   k_partial_positions  point j < n: the caller's random cell; point n + j: cell occ_list[min(floor(u[j] * K), K - 1)] (u uniform in
//                  [0,1): the same distribution as randint(K)), coordinates by Morton inversion; then k_positions' arithmetic.
//                  K == 0 (the reference then keeps the n random points only): index -1, which k_scatter_max skips.
// No host read anywhere: 15 of every 16 maintenance calls of a long run are partial sweeps.  The order of the 2n points does not
// matter to the result (scatter-max, then EMA), so the production path draws both halves SORTED (below): cells in Morton order
// make neighbouring points share table lines -- the encoder took 388 us for 1 M shuffled points, 232 us for sorted ones -- and
// neighbouring atomics share cache lines (k_scatter_max 42 -> 17 us).
constexpr uint32_t OCC_CHUNK = 1024;
__device__ __forceinline__ uint32_t compact_bits(uint32_t v) {          // inverse of expand_bits (raymarching.cu:68-75)
    v &= 0x49249249u;
    v = (v ^ (v >> 2)) & 0xC30C30C3u;
    v = (v ^ (v >> 4)) & 0x0F00F00Fu;
    v = (v ^ (v >> 8)) & 0xFF0000FFu;
    v = (v ^ (v >> 16)) & 0x0000FFFFu;
    return v;
}
// Sorted i.i.d. uniforms without a sort: the ascending order statistics of n uniforms are S_1 / S_{n+1}, ..., S_n / S_{n+1}
// with S_k the partial sums of n + 1 independent Exp(1) variables.  The caller supplies the uniforms rnd[0 .. n] the exponentials
// are made of (e = -log1p(-rnd), double); blocks of 256 terms are summed here, scanned by k_occ_scan and finished by
// k_partial_positions.  Two sequences per call (random cells, occupied ranks).
constexpr uint32_t EXP_CHUNK = 256;
__device__ __forceinline__ double exp1_of(float r) { return -log1p(-(double)r); }
__device__ __forceinline__ double block_incl_scan_f64(double v, double* total, double* lds /*5*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const double o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    double base = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) base += i < w ? lds[i] : 0.0;
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc;
}
// blocks [0, n_chunks): occupied cells per 1024-cell chunk; blocks [n_chunks, n_chunks + 2 m_chunks): sums of 256 exponentials
__global__ __launch_bounds__(256) void k_occ_count(const float* __restrict__ grid, uint32_t cells, uint32_t* __restrict__ counts,
                                                   uint32_t n_chunks, const float* __restrict__ rnd, uint32_t n1, uint32_t m_chunks,
                                                   double* __restrict__ esums) {
    __shared__ uint32_t red[4];
    __shared__ double dlds[5];
    if (blockIdx.x >= n_chunks) {
        const uint32_t q = blockIdx.x - n_chunks, seq = q / m_chunks, ch = q - seq * m_chunks;
        const uint32_t i = ch * EXP_CHUNK + threadIdx.x;
        double tot;
        (void)block_incl_scan_f64(i < n1 ? exp1_of(rnd[(size_t)seq * n1 + i]) : 0.0, &tot, dlds);
        if (threadIdx.x == 0) esums[q] = tot;
        return;
    }
    uint32_t c = 0;
    const uint32_t base = blockIdx.x * OCC_CHUNK;
#pragma unroll
    for (uint32_t k = 0; k < OCC_CHUNK / 256; k++) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        c += (i < cells && grid[i] > 0.0f) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one workgroup: counts -> exclusive prefix (+ total), and the two sequences' block sums -> exclusive prefixes (+ totals)
__global__ __launch_bounds__(1024) void k_occ_scan(uint32_t* __restrict__ counts, uint32_t n_chunks, uint32_t* __restrict__ total,
                                                   double* __restrict__ esums, uint32_t m_chunks, double* __restrict__ etotals) {
    __shared__ uint32_t lds[17];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_chunks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_chunks ? counts[i] : 0u;
        uint32_t tot;
        const uint32_t ex = lae::block_excl_scan<16>(v, &tot, lds);
        if (i < n_chunks) counts[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
    if (m_chunks && threadIdx.x < 128) {                                 // waves 0 and 1: one sequence each, 64 chunk sums per round
        const uint32_t seq = threadIdx.x >> 6, lane = threadIdx.x & 63;
        double* es = esums + (size_t)seq * m_chunks;
        double run = 0.0;
        for (uint32_t base = 0; base < m_chunks; base += 64) {
            const uint32_t i = base + lane;
            const double v = i < m_chunks ? es[i] : 0.0;
            double inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const double o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
            if (i < m_chunks) es[i] = run + inc - v;
            run += __shfl(inc, 63, 64);
        }
        if (lane == 0) etotals[seq] = run;
    }
}
__global__ __launch_bounds__(256) void k_occ_compact(const float* __restrict__ grid, uint32_t cells, const uint32_t* __restrict__ prefix,
                                                     int32_t* __restrict__ occ_list) {
    __shared__ uint32_t lds[5];
    const uint32_t base = blockIdx.x * OCC_CHUNK;
    uint32_t running = prefix[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < OCC_CHUNK / 256; k++) {                     // 256 consecutive cells per round: index order is kept
        const uint32_t i = base + k * 256 + threadIdx.x;
        const uint32_t occ = (i < cells && grid[i] > 0.0f) ? 1u : 0u;
        uint32_t tot;
        const uint32_t ex = lae::block_excl_scan<4>(occ, &tot, lds);
        if (occ) occ_list[running + ex] = (int32_t)i;
        running += tot;
    }
}
// rnd != NULL: both halves drawn sorted from rnd (see above; blocks [0, mb) = random cells as Morton codes floor(u * H^3), H a power
// of two; blocks [mb, 2 mb) = occupied ranks), mb = ceil(n / 256).  rnd == NULL: coords_rand / u as given by the caller.
__global__ __launch_bounds__(256) void k_partial_positions(const int32_t* __restrict__ coords_rand, const float* __restrict__ u,
                                    const float* __restrict__ rnd, const double* __restrict__ eprefix, const double* __restrict__ etotals,
                                    uint32_t m_chunks, uint32_t n, uint32_t H, uint32_t cells,
                                    float scale, float hgs, const float* __restrict__ noise, const int32_t* __restrict__ occ_list,
                                    const uint32_t* __restrict__ occ_total, float* __restrict__ xyzs, int32_t* __restrict__ indices) {
    __shared__ double dlds[5];
    const uint32_t mb = (n + 255u) / 256u;
    const uint32_t half = blockIdx.x / mb, jl = (blockIdx.x - half * mb) * 256u + threadIdx.x;      // point jl of its half
    float uj = 0.f;
    double ujd = 0.0;                                      // the sorted draw in double: fp32 cannot address every cell / rank above 2^24 (H >= 512: ADVICE r4)
    if (rnd) {
        const uint32_t n1 = n + 1u;
        double tot;
        const double inc = block_incl_scan_f64(jl < n1 ? exp1_of(rnd[(size_t)half * n1 + jl]) : 0.0, &tot, dlds);
        ujd = fmin((eprefix[(size_t)half * m_chunks + (jl / EXP_CHUNK)] + inc) / etotals[half], 0.99999999999999989);
        uj = fminf((float)ujd, 0.99999994f);
    }
    if (jl >= n) return;
    const uint32_t j = half * n + jl;
    int32_t c[3];
    bool valid = true;
    if (half == 0) {
        if (rnd) {
            const uint32_t code = min((uint32_t)(ujd * (double)cells), cells - 1u);
            c[0] = (int32_t)compact_bits(code); c[1] = (int32_t)compact_bits(code >> 1); c[2] = (int32_t)compact_bits(code >> 2);
        } else { c[0] = coords_rand[3 * (size_t)jl]; c[1] = coords_rand[3 * (size_t)jl + 1]; c[2] = coords_rand[3 * (size_t)jl + 2]; }
    } else {
        if (!rnd) uj = u[jl];
        const uint32_t K = *occ_total;
        valid = K > 0;
        uint32_t cell = 0;
        // rnd: the rank from the double draw; u (the caller's recorded fp32 draws): the reference's arithmetic (renderer.py:606)
        if (valid) cell = (uint32_t)occ_list[min(rnd ? (uint32_t)(ujd * (double)K) : (uint32_t)(uj * (float)K), K - 1u)];
        c[0] = (int32_t)compact_bits(cell); c[1] = (int32_t)compact_bits(cell >> 1); c[2] = (int32_t)compact_bits(cell >> 2);   // morton3D_invert
    }
    const float hm1 = (float)(H - 1);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float p = cell_unit(c[k], hm1) * scale;
        if (noise) p = p + (noise[3 * (size_t)j + k] * 2.0f - 1.0f) * hgs;
        xyzs[3 * (size_t)j + k] = valid ? p : 0.0f;
    }
    indices[j] = valid ? (int32_t)morton_encode((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]) : -1;
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

uint64_t lae_density_grid_partial_scratch_bytes(uint32_t cells, uint32_t n) {
    const uint64_t m_chunks = lae::cdiv((uint64_t)n + 1, EXP_CHUNK);
    // counts + total words, the occupied-cell list, then (8-byte aligned: ADVICE r4) the double sums
    return (((uint64_t)lae::cdiv(cells, OCC_CHUNK) + 64) * 4 + (uint64_t)cells * 4 + 7) / 8 * 8 + (2 * m_chunks + 8) * 8;
}

int lae_density_grid_partial_positions(const float* grid_c, uint32_t cells, const int32_t* coords_rand, const float* u, const float* rnd,
                                       uint32_t n, uint32_t H, float bound_c, const float* noise, float* xyzs, int32_t* indices,
                                       void* scratch, void* stream) {
    if (n == 0) return LAE_OK;
    if (!grid_c || !xyzs || !indices || !scratch || (!rnd && (!coords_rand || !u))) return LAE_ENULL;
    if (H < 2 || H > 1024 || (uint64_t)cells != (uint64_t)H * H * H || n > 0x3fffffffu) return LAE_EINVAL;
    if (rnd && (H & (H - 1)) != 0) return LAE_EINVAL;                   // sorted cell draws go through Morton codes
    const uint32_t n_chunks = lae::cdiv(cells, OCC_CHUNK);
    const uint32_t m_chunks = rnd ? lae::cdiv((uint64_t)n + 1, EXP_CHUNK) : 0u;
    uint32_t* counts = reinterpret_cast<uint32_t*>(scratch);            // [n_chunks] counts -> prefix, then the total
    uint32_t* total = counts + n_chunks;
    int32_t* occ_list = reinterpret_cast<int32_t*>(counts + n_chunks + 64);
    // [2][m_chunks] block sums -> prefixes, then the two totals; n_chunks + 64 + cells words can be an ODD count (H = 2, 4, 8):
    // the doubles start at the next 8-byte boundary (lae_density_grid_partial_scratch_bytes carries the pad)
    double* esums = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(occ_list + cells) + 7u) & ~uintptr_t(7));
    double* etotals = esums + 2 * (size_t)lae::cdiv((uint64_t)n + 1, EXP_CHUNK);
    hipStream_t s = STREAM(stream);
    k_occ_count<<<n_chunks + 2 * m_chunks, 256, 0, s>>>(grid_c, cells, counts, n_chunks, rnd, n + 1, m_chunks, esums);
    k_occ_scan<<<1, 1024, 0, s>>>(counts, n_chunks, total, esums, m_chunks, etotals);
    k_occ_compact<<<n_chunks, 256, 0, s>>>(grid_c, cells, counts, occ_list);
    const float hgs = bound_c / (float)H;
    k_partial_positions<<<2 * lae::cdiv(n, 256), 256, 0, s>>>(coords_rand, u, rnd, esums, etotals, m_chunks, n, H, cells, bound_c - hgs, hgs, noise,
                                                             occ_list, total, xyzs, indices);
    return lae::check_launch("density_grid_partial_positions");
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
