// palette.hip -- LAENeRF's palette recomposition (editing/style_encoder.py:135-158, forward_train / forward) as one
// kernel per direction.
//
//   w_hat = softmax(weight_net(x)[:, active])            fp32 (torch autocasts softmax to float32)
//   o_hat = tanh(offset_net(cat(x, SH3(d))))             fp16
//   pred  = clamp(w_hat @ palette[active].half() + o_hat, 0, 1)     fp16 (autocast matmul)
//
// In torch that is slice, softmax, tanh, cast, matmul, add, clamp plus their backward nodes (~20 launches over [P, <=16]
// tensors).  One lane per point here; the palette gradient is reduced per workgroup into a slab and summed in fixed
// order (deterministic).  Inputs are the two MLP outputs as the fused MLP writes them: [M,16] fp16, padded columns.
#include "lae_common.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
constexpr int PAL_MAX = 16;          // FFMLP output width
constexpr int PAL_BLOCK = 256;

// Every per-base array below is indexed by the base number k with COMPILE-TIME indices (loops fully unrolled, inactive bases
// predicated off by the wave-uniform mask).  Rounds 1-3 compacted the active bases into arrays indexed by a running count: a
// run-time register index, so the arrays lived in scratch memory (196 / 400 bytes per lane) and the forward / backward kernels
// took 23 / 39 us for 100 k points (profiles/r4_style_kernel_stats.csv).  Same operations in the same order: same bits.
struct Palette { float c[PAL_MAX][3]; };   // row k = base k rounded to fp16 values (palette.half()), zeros for inactive bases

__device__ __forceinline__ bool base_active(uint32_t k, uint32_t P, uint32_t mask) { return k < P && ((mask >> k) & 1u); }
__device__ __forceinline__ uint32_t compact_col(uint32_t k, uint32_t mask) { return (uint32_t)__popc(mask & ((1u << k) - 1u)); }   // column of base k among the active ones

// palette [P,3] fp32 in device memory (uniform address -> scalar loads)
__device__ __forceinline__ Palette load_palette(const float* __restrict__ palette, uint32_t P, uint32_t mask) {
    Palette pal;
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) {
        const bool act = base_active(k, P, mask);
#pragma unroll
        for (int c = 0; c < 3; c++) pal.c[k][c] = act ? (float)(half_t)palette[k * 3 + c] : 0.0f;
    }
    return pal;
}

struct Row16 { half_t v[16]; };

__device__ __forceinline__ Row16 load_row16(const half_t* __restrict__ p) {
    Row16 r;
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
    *reinterpret_cast<uint4*>(&r.v[0]) = a;
    *reinterpret_cast<uint4*>(&r.v[8]) = b;
    return r;
}

// softmax over the active columns of one row (fp32, max-subtracted like torch); w[k] = 0 for inactive bases
__device__ __forceinline__ void softmax_active(const Row16& r, uint32_t P, uint32_t mask, float (&w)[PAL_MAX]) {
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) if (base_active(k, P, mask)) mx = fmaxf(mx, (float)r.v[k]);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) {
        w[k] = 0.0f;
        if (base_active(k, P, mask)) { w[k] = expf((float)r.v[k] - mx); sum += w[k]; }
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) if (base_active(k, P, mask)) w[k] *= inv;
}

__device__ __forceinline__ void recompose(const float (&w)[PAL_MAX], uint32_t P, uint32_t mask, const Palette& pal, const half_t (&o)[3],
                                          half_t (&pre)[3]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < PAL_MAX; k++)
            if (base_active(k, P, mask)) acc = fmaf((float)(half_t)w[k], pal.c[k][c], acc);            // half operands, fp32 accumulate
        pre[c] = (half_t)((float)(half_t)acc + (float)o[c]);                                            // half matmul result + half offset
    }
}

__global__ __launch_bounds__(PAL_BLOCK) void k_palette_fwd(const half_t* __restrict__ w_logits, const half_t* __restrict__ o_raw,
                                                           const float* __restrict__ palette, uint32_t P, uint32_t mask, uint32_t M,
                                                           half_t* __restrict__ pred, float* __restrict__ w_hat,
                                                           half_t* __restrict__ o_hat) {
    const uint32_t i = blockIdx.x * PAL_BLOCK + threadIdx.x;
    if (i >= M) return;
    const Palette pal = load_palette(palette, P, mask);
    const Row16 wl = load_row16(w_logits + (size_t)i * 16), ol = load_row16(o_raw + (size_t)i * 16);
    float w[PAL_MAX];
    softmax_active(wl, P, mask, w);
    const uint32_t na = (uint32_t)__popc(mask & ((P >= 32 ? 0u : (1u << P)) - 1u));
    half_t o[3], pre[3];
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = (half_t)tanhf((float)ol.v[c]);
    recompose(w, P, mask, pal, o, pre);
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) if (base_active(k, P, mask)) w_hat[(size_t)i * na + compact_col(k, mask)] = w[k];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o_hat[(size_t)i * 3 + c] = o[c];
        pred[(size_t)i * 3 + c] = (half_t)fminf(fmaxf((float)pre[c], 0.0f), 1.0f);
    }
}

// Point-wise losses of train_LAENeRF_step (nerf/utils.py:990-996) fused behind the recomposition:
//   loss = MSE(pred, target) + w_uniform * max_j sum_i w_ij + w_non_uniform * sum_i (1 - max_j w_ij) + c_offset * sum o^2
// (style_encoder.py:183-205).  `fin` is the result block of k_style_loss_final (totals, arg-max column, loss scale).
struct StyleLossW { float w_uniform, w_non_uniform, c_offset; };
struct LossSrc { const float* target; const float* fin; const float* upstream; StyleLossW lw; };
constexpr int SL_COLS = 3 + PAL_MAX;             // partial sums per workgroup: squared error, o^2, 1 - max w, column sums
constexpr int FIN_LOSS_SCALED = 0, FIN_LOSS = 1, FIN_MSE = 2, FIN_UNIFORM = 3, FIN_NON_UNIFORM = 4, FIN_OFFSET = 5, FIN_JMAX = 6,
              FIN_SCALE = 7, FIN_REG = 8;      // fin: 12 floats

// g_pred / g_o: fp16 [M,3] or NULL; g_w: fp32 [M, n_active] or NULL (LOSS: derived from the fused criterion instead).
// Writes g_wl, g_ol [M,16] fp16 (zeros in padded / inactive columns) and this workgroup's palette-gradient partial into
// slab[blockIdx][PAL_MAX*3].
template <bool LOSS>
__global__ __launch_bounds__(PAL_BLOCK) void k_palette_bwd(const half_t* __restrict__ w_logits, const half_t* __restrict__ o_raw,
                                                           const float* __restrict__ palette, uint32_t P, uint32_t mask, uint32_t M,
                                                           const half_t* __restrict__ g_pred, const float* __restrict__ g_w,
                                                           const half_t* __restrict__ g_o, half_t* __restrict__ g_wl,
                                                           half_t* __restrict__ g_ol, float* __restrict__ slab, LossSrc ls) {
    __shared__ float red[PAL_BLOCK / 64][PAL_MAX * 3];
    const uint32_t i = blockIdx.x * PAL_BLOCK + threadIdx.x;
    const Palette pal = load_palette(palette, P, mask);
    float gp_pal[PAL_MAX][3];
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) { gp_pal[k][0] = 0.f; gp_pal[k][1] = 0.f; gp_pal[k][2] = 0.f; }
    if (i < M) {
        const Row16 wl = load_row16(w_logits + (size_t)i * 16), ol = load_row16(o_raw + (size_t)i * 16);
        float w[PAL_MAX];
        softmax_active(wl, P, mask, w);
        half_t o[3], pre[3];
#pragma unroll
        for (int c = 0; c < 3; c++) o[c] = (half_t)tanhf((float)ol.v[c]);
        recompose(w, P, mask, pal, o, pre);
        float gpc[3], got[3];
        float gmul = 0.0f;                                   // LOSS: upstream d(loss) x loss scale
        uint32_t jmax_col = 0;
        float wmax = -1.0f;
        int kmax_row = -1;
        if constexpr (LOSS) {
            gmul = ls.upstream[0] * ls.fin[FIN_SCALE];
            jmax_col = (uint32_t)ls.fin[FIN_JMAX];
#pragma unroll
            for (int k = 0; k < PAL_MAX; k++)                                                           // first maximum, like torch.max
                if (base_active(k, P, mask) && (kmax_row < 0 || w[k] > wmax)) { wmax = w[k]; kmax_row = k; }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float p = (float)pre[c];
            const bool pass = p >= 0.0f && p <= 1.0f;                                                   // clamp backward (inclusive)
            if constexpr (LOSS) {
                const float pc = fminf(fmaxf(p, 0.0f), 1.0f);
                gpc[c] = pass ? gmul * 2.0f * (pc - ls.target[(size_t)i * 3 + c]) / (3.0f * (float)M) : 0.0f;
                got[c] = gpc[c] + gmul * 2.0f * ls.lw.c_offset * (float)o[c];
            } else {
                gpc[c] = (g_pred && pass) ? (float)g_pred[(size_t)i * 3 + c] : 0.0f;
                got[c] = gpc[c] + (g_o ? (float)g_o[(size_t)i * 3 + c] : 0.0f);
            }
        }
        const uint32_t na = (uint32_t)__popc(mask & ((P >= 32 ? 0u : (1u << P)) - 1u));
        float gw[PAL_MAX], dot = 0.0f;
#pragma unroll
        for (int k = 0; k < PAL_MAX; k++) {
            gw[k] = 0.0f;
            if (base_active(k, P, mask)) {
                float gin;
                if constexpr (LOSS) gin = gmul * ((compact_col(k, mask) == jmax_col ? ls.lw.w_uniform : 0.0f) - (k == kmax_row ? ls.lw.w_non_uniform : 0.0f));
                else gin = g_w ? g_w[(size_t)i * na + compact_col(k, mask)] : 0.0f;
                gw[k] = gin + gpc[0] * pal.c[k][0] + gpc[1] * pal.c[k][1] + gpc[2] * pal.c[k][2];
                dot = fmaf(w[k], gw[k], dot);
                gp_pal[k][0] = w[k] * gpc[0]; gp_pal[k][1] = w[k] * gpc[1]; gp_pal[k][2] = w[k] * gpc[2];
            }
        }
        Row16 out_w, out_o;
#pragma unroll
        for (int k = 0; k < 16; k++) { out_w.v[k] = (half_t)0.0f; out_o.v[k] = (half_t)0.0f; }
#pragma unroll
        for (int k = 0; k < PAL_MAX; k++)
            if (base_active(k, P, mask)) out_w.v[k] = (half_t)(w[k] * (gw[k] - dot));                      // softmax backward
#pragma unroll
        for (int c = 0; c < 3; c++) { const float t = (float)o[c]; out_o.v[c] = (half_t)(got[c] * (1.0f - t * t)); }   // tanh backward
        *reinterpret_cast<uint4*>(g_wl + (size_t)i * 16) = *reinterpret_cast<const uint4*>(&out_w.v[0]);
        *reinterpret_cast<uint4*>(g_wl + (size_t)i * 16 + 8) = *reinterpret_cast<const uint4*>(&out_w.v[8]);
        *reinterpret_cast<uint4*>(g_ol + (size_t)i * 16) = *reinterpret_cast<const uint4*>(&out_o.v[0]);
        *reinterpret_cast<uint4*>(g_ol + (size_t)i * 16 + 8) = *reinterpret_cast<const uint4*>(&out_o.v[8]);
    }
    // palette gradient: wave reduce of the active bases (uniform mask) -> LDS (compact rows) -> one partial row per workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < PAL_BLOCK / 64 * PAL_MAX * 3) (&red[0][0])[threadIdx.x] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) {
        if (base_active(k, P, mask)) {
            const uint32_t j = compact_col(k, mask);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float v = gp_pal[k][c];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                if (lane == 0) red[wv][j * 3 + c] = v;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < PAL_MAX * 3) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < PAL_BLOCK / 64; q++) t += red[q][threadIdx.x];
        slab[(size_t)blockIdx.x * (PAL_MAX * 3) + threadIdx.x] = t;
    }
}

// The palette-only regulariser of train_LAENeRF_step (style_encoder.py:195-202, `palet_loss`): over ALL P bases
//   valid = sum floor(p) * p ;  dists_ij = |p_i - p_j|^2, m = max dists ;  distinct = mean_ij (1 - dists_ij / m)
//   reg = w_valid * valid + w_distinct * distinct.
// In torch: ~20 tiny kernels forward and as many backward on a [P,3] tensor, every step.  Here the value is one thread's work in
// k_style_loss_final and the gradient one thread's per palette entry in k_palette_grad_reduce.  d(max): torch's full-reduction max
// spreads the gradient evenly over tied maxima; dists is symmetric, so there are always at least two.
struct PalReg { const float* palette; uint32_t P; float w_valid, w_distinct; };
__device__ __forceinline__ float pal_dist(const float* __restrict__ pal, uint32_t i, uint32_t j) {
    float d = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) { const float t = pal[i * 3 + c] - pal[j * 3 + c]; d += t * t; }
    return d;
}
// wave-cooperative (all 64 lanes of one wave call these with the same arguments; results are wave-uniform): lane q takes the pairs
// (i, j) = (q / P, q % P), q + 64, ... -- one thread looping over the P^2 pairs with dependent LDS reads took 13 us
struct PalStats { float S, m, ties; };
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ PalStats pal_reg_stats(const PalReg& r, int lane) {
    float S = 0.0f, m = 0.0f, ties = 0.0f;
    for (uint32_t q = (uint32_t)lane; q < r.P * r.P; q += 64u) { const float d = pal_dist(r.palette, q / r.P, q % r.P); S += d; m = fmaxf(m, d); }
    S = wave_sum(S);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    for (uint32_t q = (uint32_t)lane; q < r.P * r.P; q += 64u) ties += pal_dist(r.palette, q / r.P, q % r.P) == m ? 1.0f : 0.0f;
    return PalStats{S, m, wave_sum(ties)};
}
__device__ __forceinline__ float pal_reg_value(const PalReg& r, int lane) {
    const PalStats st = pal_reg_stats(r, lane);
    float valid = 0.0f;
    for (uint32_t q = (uint32_t)lane; q < r.P * 3u; q += 64u) valid += floorf(r.palette[q]) * r.palette[q];
    valid = wave_sum(valid);
    return r.w_valid * valid + r.w_distinct * (1.0f - st.S / ((float)(r.P * r.P) * st.m));
}
__device__ __forceinline__ float pal_reg_grad(const PalReg& r, const PalStats& st, uint32_t k, uint32_t c, int lane) {
    float dm = 0.0f, dS = 0.0f;
    for (uint32_t q = (uint32_t)lane; q < r.P * r.P; q += 64u) {
        const uint32_t i = q / r.P, j = q % r.P;
        if (pal_dist(r.palette, i, j) == st.m) dm += 2.0f * (r.palette[i * 3 + c] - r.palette[j * 3 + c]) * ((i == k ? 1.0f : 0.0f) - (j == k ? 1.0f : 0.0f));
    }
    for (uint32_t j = (uint32_t)lane; j < r.P; j += 64u) dS += 4.0f * (r.palette[k * 3 + c] - r.palette[j * 3 + c]);
    dm = wave_sum(dm) / st.ties; dS = wave_sum(dS);
    return r.w_valid * floorf(r.palette[k * 3 + c]) - r.w_distinct * (dS / st.m - st.S / (st.m * st.m) * dm) / (float)(r.P * r.P);
}

// fixed-order sum of the slabs (one workgroup per (compact row j, channel c): lanes stride over the partials, then a
// butterfly -- the same order every run); scatter the compact active rows back to the [P,3] parameter gradient; reg.palette != NULL:
// + gmul[0] * gmul[1] * d(reg) / d(palette) on every entry (gmul = upstream, fin: the scaled upstream gradient of the criterion)
__global__ __launch_bounds__(64) void k_palette_grad_reduce(const float* __restrict__ slab, uint32_t n_blocks, uint32_t P, uint32_t mask,
                                                            float* __restrict__ g_palette, PalReg reg, const float* __restrict__ upstream,
                                                            const float* __restrict__ fin, int accumulate) {
    __shared__ float spal[PAL_MAX * 3];                  // the regulariser's loops read the palette ~P^2 times: from LDS, not from memory
    if (reg.palette && threadIdx.x < reg.P * 3u) spal[threadIdx.x] = reg.palette[threadIdx.x];
    __syncthreads();
    if (reg.palette) reg.palette = spal;
    const uint32_t e = blockIdx.x;                       // (compact row j, channel c)
    float t = 0.0f;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 64) t += slab[(size_t)b * (PAL_MAX * 3) + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
    const uint32_t j = e / 3, c = e % 3;
    const float gmul = reg.palette ? upstream[0] * fin[7] : 0.0f;       // fin[FIN_SCALE]
    PalStats st{0.f, 1.f, 1.f};
    if (reg.palette) st = pal_reg_stats(reg, (int)threadIdx.x);
    uint32_t seen = 0;
    for (uint32_t k = 0; k < P; k++) {                   // uniform control flow: every lane takes part in the regulariser's sums
        const bool act = (mask >> k) & 1u;
        const bool mine = act ? seen == j : j == 0;
        if (mine) {
            const float g = reg.palette ? gmul * pal_reg_grad(reg, st, k, c, (int)threadIdx.x) : 0.0f;
            if (threadIdx.x == 0) {
                const float v = (act ? t : 0.0f) + g;
                g_palette[k * 3 + c] = accumulate ? g_palette[k * 3 + c] + v : v;    // accumulate: the caller's persistent .grad (fp32 add, like AccumulateGrad)
            }
        }
        seen += act ? 1u : 0u;
    }
}

// forward of the fused criterion: per-workgroup partial sums (fixed order inside the wave / workgroup)
__global__ __launch_bounds__(PAL_BLOCK) void k_style_loss_partial(const half_t* __restrict__ pred, const float* __restrict__ target,
                                                                  const float* __restrict__ w_hat, const half_t* __restrict__ o_hat,
                                                                  uint32_t M, uint32_t na, float* __restrict__ slab) {
    __shared__ float red[PAL_BLOCK / 64][SL_COLS];
    const uint32_t i = blockIdx.x * PAL_BLOCK + threadIdx.x;
    float v[SL_COLS];
#pragma unroll
    for (int k = 0; k < SL_COLS; k++) v[k] = 0.0f;
    if (i < M) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = (float)pred[(size_t)i * 3 + c] - target[(size_t)i * 3 + c];
            const float o = (float)o_hat[(size_t)i * 3 + c];
            v[0] = fmaf(d, d, v[0]); v[1] = fmaf(o, o, v[1]);
        }
        float mx = -1.0f;
#pragma unroll
        for (int j = 0; j < PAL_MAX; j++)
            if ((uint32_t)j < na) { const float w = w_hat[(size_t)i * na + j]; v[3 + j] = w; mx = fmaxf(mx, w); }
        v[2] = 1.0f - mx;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < SL_COLS; k++) {
        float t = v[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (lane == 0) red[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < SL_COLS) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < PAL_BLOCK / 64; q++) t += red[q][threadIdx.x];
        slab[(size_t)blockIdx.x * SL_COLS + threadIdx.x] = t;
    }
}

// one workgroup: fixed-order totals of the partials, the loss terms, the arg-max column of the uniform term
__global__ __launch_bounds__(1024) void k_style_loss_final(const float* __restrict__ slab, uint32_t n_blocks, uint32_t M, uint32_t na,
                                                           StyleLossW lw, const float* __restrict__ scale, float* __restrict__ fin, PalReg reg) {
    __shared__ float tot[SL_COLS];
    __shared__ float spal[PAL_MAX * 3];
    if (reg.palette && threadIdx.x < reg.P * 3u) spal[threadIdx.x] = reg.palette[threadIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int col = wave; col < SL_COLS; col += 16) {         // a wave per column: lanes stride over the partials, then a butterfly
        float t = 0.0f;
        for (uint32_t b = lane; b < n_blocks; b += 64) t += slab[(size_t)b * SL_COLS + col];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (lane == 0) tot[col] = t;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (reg.palette) reg.palette = spal;                                 // (written before the barrier above)
    const float regv = reg.palette ? pal_reg_value(reg, (int)threadIdx.x) : 0.0f;      // wave 0, all lanes
    if (threadIdx.x != 0) return;
    uint32_t jmax = 0;
    for (uint32_t j = 1; j < na; j++) if (tot[3 + j] > tot[3 + jmax]) jmax = j;
    const float s = scale ? scale[0] : 1.0f;
    const float mse = tot[0] / (3.0f * (float)M), uni = lw.w_uniform * tot[3 + jmax], non = lw.w_non_uniform * tot[2],
                off = lw.c_offset * tot[1];
    // nerf/utils.py:990-995: the criterion's fp32 value, then `loss += weights_loss(...).half()`, `+= offset_loss(...).half()`,
    // `+= palet_loss(params).half()` -- each added term rounded to fp16 first (the casts' backward is the identity, so only the
    // reported / scaled VALUE sees it; fin[] keeps the unrounded terms).  ADVICE r4.
    const auto h = [](float v) { return (float)(half_t)v; };
    const float loss = ((mse + h(uni + non)) + h(off)) + h(regv);
    fin[FIN_REG] = regv;
    fin[FIN_LOSS_SCALED] = loss * s; fin[FIN_LOSS] = loss; fin[FIN_MSE] = mse; fin[FIN_UNIFORM] = uni; fin[FIN_NON_UNIFORM] = non;
    fin[FIN_OFFSET] = off; fin[FIN_JMAX] = (float)jmax; fin[FIN_SCALE] = s;
}

int check_palette(uint32_t P, uint32_t mask) {
    if (P == 0 || P > PAL_MAX) return LAE_EINVAL;
    return (mask & ((1u << P) - 1u)) ? LAE_OK : LAE_EINVAL;        // at least one active base
}


// ---- LAENeRF input assembly (round 5): what sits between the hash-grid encoder and the two MLPs of LAENeRF.forward_train
// (editing/style_encoder.py:135-146).  The reference takes the encoder's [M,32] rows, evaluates SH(3) of the directions, casts,
// pads and concatenates: with this repo's operators that was a transpose of the encoder's level-major output, an SH launch, a
// division by `size`, a cast, a zero fill and a cat (31 us, six launches per step of 100 k points) and, backwards, two slice
// copies, an add and the inverse transpose (20 us, four launches).  Here: ONE kernel each way.  The values are the separate
// operators' bit for bit (same SH evaluation, same fp16 roundings, the two gradients added in fp32 and rounded once like torch's
// half add).
#include "sh_table.inc"
constexpr int SA_BLOCK = 256;
// feats_lm [16][M] half2 (level-major) + dirs [M,3] -> feat [Mp,32] half rows and off_in [Mp,off_cols] half rows = [feat | SH(DEG) | 0];
// rows M..Mp-1 (the MLPs want a multiple of 16 rows) are zero.  DEG = 0: no directions, off_in is not written.
template <int DEG>
__global__ __launch_bounds__(SA_BLOCK) void k_style_assemble_fwd(const uint32_t* __restrict__ feats_lm, const float* __restrict__ dirs, uint32_t M,
                                                                 uint32_t Mp, uint32_t* __restrict__ feat, uint32_t* __restrict__ off_in,
                                                                 uint32_t off_words) {
    constexpr int C2 = DEG * DEG;
    constexpr int MAXW = 24;                                // 48 halves
    __shared__ uint32_t tile[SA_BLOCK * (MAXW + 1)];
    const uint32_t base = blockIdx.x * SA_BLOCK, b = base + threadIdx.x;
    uint32_t* row = tile + threadIdx.x * (MAXW + 1);
    const bool live = b < M;
#pragma unroll
    for (int l = 0; l < 16; l++) row[l] = live ? feats_lm[(size_t)l * M + b] : 0u;
    if constexpr (DEG > 0) {
        float o[C2 > 0 ? C2 : 1], g0[1], g1[1], g2[1];
        float x = 0.f, y = 0.f, z = 0.f;
        if (live) { x = dirs[3 * (size_t)b]; y = dirs[3 * (size_t)b + 1]; z = dirs[3 * (size_t)b + 2]; }
        sh_eval<DEG, false>(x, y, z, o, g0, g1, g2);
        // the basis value is ROUNDED TO fp32 FIRST, like the SH operator's output that torch then casts: without the barrier the
        // compiler folds multiply + conversion into v_fma_mixlo_f16 (one rounding), which differs from the operator chain at fp32 ties
        // (1 value in 36 864 in tests/test_gpu_style.py::test_fused_input_assembly_equals_the_operator_chain)
        half_t hv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            float v = (i < C2 && live) ? o[i < C2 ? i : 0] : 0.0f;
            asm volatile("" : "+v"(v));
            hv[i] = (half_t)v;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const half2_t h2 = {hv[2 * i], hv[2 * i + 1]};
            row[16 + i] = __builtin_bit_cast(uint32_t, h2);
        }
    }
    __syncthreads();
    const uint32_t nrow = min((uint32_t)SA_BLOCK, Mp > base ? Mp - base : 0u);
    // fixed trip counts, every LDS read issued before the first store waits for it (a loop of unknown length with a division
    // by `off_words` inside was 40 % of this kernel)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t e = k * SA_BLOCK + threadIdx.x, r = e >> 4, c = e & 15u;
        if (r < nrow) feat[(size_t)(base + r) * 16u + c] = tile[r * (MAXW + 1) + c];
    }
    if constexpr (DEG > 0) {
        const uint32_t inv = (uint32_t)((0x100000000ull + off_words - 1u) / off_words);      // e / off_words == umulhi(e, inv) for e < 2^16
        const uint32_t total = nrow * off_words;
#pragma unroll
        for (int k = 0; k < MAXW; k++) {
            const uint32_t e = k * SA_BLOCK + threadIdx.x, r = __umulhi(e, inv), c = e - r * off_words;
            if (e < total) off_in[(size_t)(base + r) * off_words + c] = tile[r * (MAXW + 1) + c];
        }
    }
}
// grad_lm [16][M] half2 = g_feat [Mp,32] rows + the first 32 columns of g_off [Mp,off_cols] rows (either may be NULL)
__global__ __launch_bounds__(SA_BLOCK) void k_style_assemble_bwd(const uint32_t* __restrict__ g_feat, const uint32_t* __restrict__ g_off, uint32_t M,
                                                                 uint32_t off_words, uint32_t* __restrict__ grad_lm) {
    __shared__ uint32_t tile[SA_BLOCK * 17];
    const uint32_t base = blockIdx.x * SA_BLOCK;
    const uint32_t nrow = min((uint32_t)SA_BLOCK, M - base);
    // 16 words per thread, all loads issued before the first use (a loop of unknown trip count serialises the load latencies:
    // 14.0 -> see profiles/r5c_style_kernel_stats.csv)
    uint32_t vf[16], vo[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t e = k * SA_BLOCK + threadIdx.x, r = e >> 4, c = e & 15u;
        const bool in = r < nrow;
        vf[k] = (g_feat && in) ? g_feat[(size_t)(base + r) * 16u + c] : 0u;
        vo[k] = (g_off && in) ? g_off[(size_t)(base + r) * off_words + c] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t e = k * SA_BLOCK + threadIdx.x, r = e >> 4, c = e & 15u;
        const half2_t a = __builtin_bit_cast(half2_t, vf[k]), b2 = __builtin_bit_cast(half2_t, vo[k]);
        half2_t s2;
        if (g_feat && g_off) s2 = half2_t{(half_t)((float)a[0] + (float)b2[0]), (half_t)((float)a[1] + (float)b2[1])};
        else s2 = g_feat ? a : b2;
        tile[r * 17 + c] = __builtin_bit_cast(uint32_t, s2);
    }
    __syncthreads();
    const uint32_t b = base + threadIdx.x;
    if (b >= M) return;
#pragma unroll
    for (int l = 0; l < 16; l++) grad_lm[(size_t)l * M + b] = tile[threadIdx.x * 17 + l];
}

}  // namespace

extern "C" {

int lae_style_assemble_forward(const void* feats_lm, const float* dirs, uint32_t M, uint32_t Mp, uint32_t degree, void* feat, void* off_in,
                               uint32_t off_cols, void* stream) {
    if (Mp == 0) return LAE_OK;
    if (!feats_lm || !feat || (degree && (!dirs || !off_in))) return LAE_ENULL;
    if (Mp < M || degree > 4 || (degree && (off_cols % 2u || off_cols > 48u || off_cols < 32u + degree * degree))) return LAE_EINVAL;
    const uint32_t nb = lae::cdiv(Mp, (uint32_t)SA_BLOCK), ow = off_cols / 2u;
    hipStream_t s = STREAM(stream);
    const uint32_t* f = reinterpret_cast<const uint32_t*>(feats_lm);
    uint32_t *fo = reinterpret_cast<uint32_t*>(feat), *oo = reinterpret_cast<uint32_t*>(off_in);
    switch (degree) {
        case 0: k_style_assemble_fwd<0><<<nb, SA_BLOCK, 0, s>>>(f, nullptr, M, Mp, fo, nullptr, 0u); break;
        case 1: k_style_assemble_fwd<1><<<nb, SA_BLOCK, 0, s>>>(f, dirs, M, Mp, fo, oo, ow); break;
        case 2: k_style_assemble_fwd<2><<<nb, SA_BLOCK, 0, s>>>(f, dirs, M, Mp, fo, oo, ow); break;
        case 3: k_style_assemble_fwd<3><<<nb, SA_BLOCK, 0, s>>>(f, dirs, M, Mp, fo, oo, ow); break;
        default: k_style_assemble_fwd<4><<<nb, SA_BLOCK, 0, s>>>(f, dirs, M, Mp, fo, oo, ow); break;
    }
    return lae::check_launch("style_assemble_forward");
}

int lae_style_assemble_backward(const void* g_feat, const void* g_off, uint32_t M, uint32_t off_cols, void* grad_lm, void* stream) {
    if (M == 0) return LAE_OK;
    if (!grad_lm || (!g_feat && !g_off)) return LAE_ENULL;
    if (g_off && (off_cols % 2u || off_cols < 32u || off_cols > 48u)) return LAE_EINVAL;
    k_style_assemble_bwd<<<lae::cdiv(M, (uint32_t)SA_BLOCK), SA_BLOCK, 0, STREAM(stream)>>>(
        reinterpret_cast<const uint32_t*>(g_feat), reinterpret_cast<const uint32_t*>(g_off), M, off_cols / 2u, reinterpret_cast<uint32_t*>(grad_lm));
    return lae::check_launch("style_assemble_backward");
}

int lae_palette_forward(const void* w_logits, const void* o_raw, const float* palette, uint32_t P, uint32_t active_mask, uint32_t M,
                        void* pred, float* w_hat, void* o_hat, void* stream) {
    if (M == 0) return LAE_OK;
    if (!w_logits || !o_raw || !palette || !pred || !w_hat || !o_hat) return LAE_ENULL;
    const int rc = check_palette(P, active_mask);
    if (rc) return rc;
    k_palette_fwd<<<lae::cdiv(M, PAL_BLOCK), PAL_BLOCK, 0, STREAM(stream)>>>((const half_t*)w_logits, (const half_t*)o_raw, palette, P,
                                                                          active_mask, M, (half_t*)pred, w_hat, (half_t*)o_hat);
    return lae::check_launch("palette_forward");
}

uint64_t lae_palette_backward_scratch_bytes(uint32_t M) { return (uint64_t)lae::cdiv(M, PAL_BLOCK) * PAL_MAX * 3 * sizeof(float) + 256; }

int lae_palette_backward(const void* w_logits, const void* o_raw, const float* palette, uint32_t P, uint32_t active_mask, uint32_t M,
                         const void* g_pred, const float* g_w, const void* g_o, void* g_w_logits, void* g_o_raw, float* g_palette,
                         void* scratch, void* stream) {
    if (!g_palette) return LAE_ENULL;
    hipStream_t s = STREAM(stream);
    if (M == 0) return hipMemsetAsync(g_palette, 0, (size_t)P * 3 * sizeof(float), s) == hipSuccess ? LAE_OK : LAE_ELAUNCH;
    if (!w_logits || !o_raw || !palette || !g_w_logits || !g_o_raw || !scratch) return LAE_ENULL;
    const int rc = check_palette(P, active_mask);
    if (rc) return rc;
    const uint32_t nb = lae::cdiv(M, PAL_BLOCK);
    k_palette_bwd<false><<<nb, PAL_BLOCK, 0, s>>>((const half_t*)w_logits, (const half_t*)o_raw, palette, P, active_mask, M,
                                                  (const half_t*)g_pred, g_w, (const half_t*)g_o, (half_t*)g_w_logits, (half_t*)g_o_raw,
                                                  (float*)scratch, LossSrc{});
    k_palette_grad_reduce<<<PAL_MAX * 3, 64, 0, s>>>((const float*)scratch, nb, P, active_mask, g_palette, PalReg{nullptr, 0, 0.f, 0.f}, nullptr, nullptr, 0);
    return lae::check_launch("palette_backward");
}

uint64_t lae_style_loss_scratch_bytes(uint32_t M) { return (uint64_t)lae::cdiv(M, PAL_BLOCK) * SL_COLS * sizeof(float) + 256; }

int lae_style_loss_forward(const void* pred, const float* target, const float* w_hat, const void* o_hat, uint32_t M, uint32_t n_active,
                           float w_uniform, float w_non_uniform, float c_offset, const float* scale, float* fin, void* scratch,
                           const float* reg_palette, uint32_t reg_P, float w_valid, float w_distinct, void* stream) {
    if (!pred || !target || !w_hat || !o_hat || !fin || !scratch) return LAE_ENULL;
    if (M == 0 || n_active == 0 || n_active > PAL_MAX || (reg_palette && (reg_P == 0 || reg_P > PAL_MAX))) return LAE_EINVAL;
    hipStream_t s = STREAM(stream);
    const uint32_t nb = lae::cdiv(M, PAL_BLOCK);
    k_style_loss_partial<<<nb, PAL_BLOCK, 0, s>>>((const half_t*)pred, target, w_hat, (const half_t*)o_hat, M, n_active, (float*)scratch);
    k_style_loss_final<<<1, 1024, 0, s>>>((const float*)scratch, nb, M, n_active, StyleLossW{w_uniform, w_non_uniform, c_offset},
                                                  scale, fin, PalReg{reg_palette, reg_P, w_valid, w_distinct});
    return lae::check_launch("style_loss_forward");
}

int lae_style_loss_backward(const void* w_logits, const void* o_raw, const float* palette, uint32_t P, uint32_t active_mask, uint32_t M,
                            const float* target, const float* fin, const float* upstream, float w_uniform, float w_non_uniform,
                            float c_offset, void* g_w_logits, void* g_o_raw, float* g_palette, void* scratch, int flags, float w_valid,
                            float w_distinct, void* stream) {
    if (!w_logits || !o_raw || !palette || !target || !fin || !upstream || !g_w_logits || !g_o_raw || !g_palette || !scratch) return LAE_ENULL;
    if (M == 0) return LAE_EINVAL;
    const int rc = check_palette(P, active_mask);
    if (rc) return rc;
    hipStream_t s = STREAM(stream);
    const uint32_t nb = lae::cdiv(M, PAL_BLOCK);
    const LossSrc ls{target, fin, upstream, StyleLossW{w_uniform, w_non_uniform, c_offset}};
    k_palette_bwd<true><<<nb, PAL_BLOCK, 0, s>>>((const half_t*)w_logits, (const half_t*)o_raw, palette, P, active_mask, M, nullptr, nullptr,
                                                 nullptr, (half_t*)g_w_logits, (half_t*)g_o_raw, (float*)scratch, ls);
    k_palette_grad_reduce<<<PAL_MAX * 3, 64, 0, s>>>((const float*)scratch, nb, P, active_mask, g_palette,
                                                     PalReg{(flags & LAE_STYLE_WITH_REG) ? palette : nullptr, P, w_valid, w_distinct}, upstream, fin,
                                                     (flags & LAE_STYLE_ACCUMULATE_PALETTE) ? 1 : 0);
    return lae::check_launch("style_loss_backward");
}

}  // extern "C"
