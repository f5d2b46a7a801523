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
constexpr int PAL_MAX = 16;          // FFMLP output width
constexpr int PAL_BLOCK = 256;

struct Palette { float c[PAL_MAX][3]; };   // active rows, compacted, rounded to fp16 values (palette.half())

// palette [P,3] fp32 in device memory (uniform address -> scalar loads); rows of inactive bases are skipped
__device__ __forceinline__ Palette load_palette(const float* __restrict__ palette, uint32_t P, uint32_t mask) {
    Palette pal;
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) { pal.c[k][0] = 0.f; pal.c[k][1] = 0.f; pal.c[k][2] = 0.f; }
    uint32_t j = 0;
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++)
        if ((uint32_t)k < P && ((mask >> k) & 1u)) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = (float)(half_t)palette[k * 3 + c];
#pragma unroll
                for (int q = 0; q < PAL_MAX; q++) if ((uint32_t)q == j) pal.c[q][c] = v;      // static register indexing
            }
            j++;
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

// softmax over the active columns of one row (fp32, max-subtracted like torch), compacted: w[0..n_active)
__device__ __forceinline__ void softmax_active(const Row16& r, uint32_t P, uint32_t mask, float (&w)[PAL_MAX], uint32_t& n_active) {
    float mx = -3.0e38f;
    n_active = 0;
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++)
        if ((uint32_t)k < P && ((mask >> k) & 1u)) { w[n_active] = (float)r.v[k]; mx = fmaxf(mx, w[n_active]); n_active++; }
    float sum = 0.0f;
    for (uint32_t j = 0; j < n_active; j++) { w[j] = expf(w[j] - mx); sum += w[j]; }
    const float inv = 1.0f / sum;
    for (uint32_t j = 0; j < n_active; j++) w[j] *= inv;
}

__device__ __forceinline__ void recompose(const float (&w)[PAL_MAX], uint32_t n_active, const Palette& pal, const half_t (&o)[3],
                                          half_t (&pre)[3]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float acc = 0.0f;
        for (uint32_t j = 0; j < n_active; j++) acc = fmaf((float)(half_t)w[j], pal.c[j][c], acc);   // half operands, fp32 accumulate
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
    uint32_t na;
    softmax_active(wl, P, mask, w, na);
    half_t o[3], pre[3];
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = (half_t)tanhf((float)ol.v[c]);
    recompose(w, na, pal, o, pre);
    for (uint32_t j = 0; j < na; j++) w_hat[(size_t)i * na + j] = w[j];
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
              FIN_SCALE = 7;

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
        uint32_t na;
        softmax_active(wl, P, mask, w, na);
        half_t o[3], pre[3];
#pragma unroll
        for (int c = 0; c < 3; c++) o[c] = (half_t)tanhf((float)ol.v[c]);
        recompose(w, na, pal, o, pre);
        float gpc[3], got[3];
        float gmul = 0.0f;                                   // LOSS: upstream d(loss) x loss scale
        uint32_t jmax_col = 0, jmax_row = 0;
        if constexpr (LOSS) {
            gmul = ls.upstream[0] * ls.fin[FIN_SCALE];
            jmax_col = (uint32_t)ls.fin[FIN_JMAX];
            for (uint32_t j = 1; j < na; j++) if (w[j] > w[jmax_row]) jmax_row = j;                     // first maximum, like torch.max
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
        float gw[PAL_MAX], dot = 0.0f;
        for (uint32_t j = 0; j < na; j++) {
            float gin;
            if constexpr (LOSS) gin = gmul * ((j == jmax_col ? ls.lw.w_uniform : 0.0f) - (j == jmax_row ? ls.lw.w_non_uniform : 0.0f));
            else gin = g_w ? g_w[(size_t)i * na + j] : 0.0f;
            gw[j] = gin + gpc[0] * pal.c[j][0] + gpc[1] * pal.c[j][1] + gpc[2] * pal.c[j][2];
            dot = fmaf(w[j], gw[j], dot);
            gp_pal[j][0] = w[j] * gpc[0]; gp_pal[j][1] = w[j] * gpc[1]; gp_pal[j][2] = w[j] * gpc[2];
        }
        Row16 out_w, out_o;
#pragma unroll
        for (int k = 0; k < 16; k++) { out_w.v[k] = (half_t)0.0f; out_o.v[k] = (half_t)0.0f; }
        uint32_t j = 0;
#pragma unroll
        for (int k = 0; k < PAL_MAX; k++)
            if ((uint32_t)k < P && ((mask >> k) & 1u)) { out_w.v[k] = (half_t)(w[j] * (gw[j] - dot)); j++; }   // softmax backward
#pragma unroll
        for (int c = 0; c < 3; c++) { const float t = (float)o[c]; out_o.v[c] = (half_t)(got[c] * (1.0f - t * t)); }   // tanh backward
        *reinterpret_cast<uint4*>(g_wl + (size_t)i * 16) = *reinterpret_cast<const uint4*>(&out_w.v[0]);
        *reinterpret_cast<uint4*>(g_wl + (size_t)i * 16 + 8) = *reinterpret_cast<const uint4*>(&out_w.v[8]);
        *reinterpret_cast<uint4*>(g_ol + (size_t)i * 16) = *reinterpret_cast<const uint4*>(&out_o.v[0]);
        *reinterpret_cast<uint4*>(g_ol + (size_t)i * 16 + 8) = *reinterpret_cast<const uint4*>(&out_o.v[8]);
    }
    // palette gradient: wave reduce (active rows only; the count is uniform) -> LDS -> one partial row per workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t na_u = (uint32_t)__popc(mask & ((P >= 32 ? 0u : (1u << P)) - 1u));
#pragma unroll
    for (int k = 0; k < PAL_MAX; k++) {
        if ((uint32_t)k < na_u) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float v = gp_pal[k][c];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                if (lane == 0) red[wv][k * 3 + c] = v;
            }
        } else if (lane == 0) { red[wv][k * 3] = 0.f; red[wv][k * 3 + 1] = 0.f; red[wv][k * 3 + 2] = 0.f; }
    }
    __syncthreads();
    if (threadIdx.x < PAL_MAX * 3) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < PAL_BLOCK / 64; q++) t += red[q][threadIdx.x];
        slab[(size_t)blockIdx.x * (PAL_MAX * 3) + threadIdx.x] = t;
    }
}

// fixed-order sum of the slabs (one workgroup per (compact row j, channel c): lanes stride over the partials, then a
// butterfly -- the same order every run); scatter the compact active rows back to the [P,3] parameter gradient
__global__ __launch_bounds__(64) void k_palette_grad_reduce(const float* __restrict__ slab, uint32_t n_blocks, uint32_t P, uint32_t mask,
                                                            float* __restrict__ g_palette) {
    const uint32_t e = blockIdx.x;                       // (compact row j, channel c)
    float t = 0.0f;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 64) t += slab[(size_t)b * (PAL_MAX * 3) + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
    if (threadIdx.x != 0) return;
    const uint32_t j = e / 3, c = e % 3;
    uint32_t seen = 0;
    for (uint32_t k = 0; k < P; k++) {
        if ((mask >> k) & 1u) { if (seen == j) g_palette[k * 3 + c] = t; seen++; }
        else if (j == 0) g_palette[k * 3 + c] = 0.0f;
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
                                                           StyleLossW lw, const float* __restrict__ scale, float* __restrict__ fin) {
    __shared__ float tot[SL_COLS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int col = wave; col < SL_COLS; col += 16) {         // a wave per column: lanes stride over the partials, then a butterfly
        float t = 0.0f;
        for (uint32_t b = lane; b < n_blocks; b += 64) t += slab[(size_t)b * SL_COLS + col];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (lane == 0) tot[col] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t jmax = 0;
    for (uint32_t j = 1; j < na; j++) if (tot[3 + j] > tot[3 + jmax]) jmax = j;
    const float s = scale ? scale[0] : 1.0f;
    const float mse = tot[0] / (3.0f * (float)M), uni = lw.w_uniform * tot[3 + jmax], non = lw.w_non_uniform * tot[2],
                off = lw.c_offset * tot[1];
    const float loss = ((mse + uni) + non) + off;
    fin[FIN_LOSS_SCALED] = loss * s; fin[FIN_LOSS] = loss; fin[FIN_MSE] = mse; fin[FIN_UNIFORM] = uni; fin[FIN_NON_UNIFORM] = non;
    fin[FIN_OFFSET] = off; fin[FIN_JMAX] = (float)jmax; fin[FIN_SCALE] = s;
}

int check_palette(uint32_t P, uint32_t mask) {
    if (P == 0 || P > PAL_MAX) return LAE_EINVAL;
    return (mask & ((1u << P) - 1u)) ? LAE_OK : LAE_EINVAL;        // at least one active base
}

}  // namespace

extern "C" {

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
    k_palette_grad_reduce<<<PAL_MAX * 3, 64, 0, s>>>((const float*)scratch, nb, P, active_mask, g_palette);
    return lae::check_launch("palette_backward");
}

uint64_t lae_style_loss_scratch_bytes(uint32_t M) { return (uint64_t)lae::cdiv(M, PAL_BLOCK) * SL_COLS * sizeof(float) + 256; }

int lae_style_loss_forward(const void* pred, const float* target, const float* w_hat, const void* o_hat, uint32_t M, uint32_t n_active,
                           float w_uniform, float w_non_uniform, float c_offset, const float* scale, float* fin, void* scratch,
                           void* stream) {
    if (!pred || !target || !w_hat || !o_hat || !fin || !scratch) return LAE_ENULL;
    if (M == 0 || n_active == 0 || n_active > PAL_MAX) return LAE_EINVAL;
    hipStream_t s = STREAM(stream);
    const uint32_t nb = lae::cdiv(M, PAL_BLOCK);
    k_style_loss_partial<<<nb, PAL_BLOCK, 0, s>>>((const half_t*)pred, target, w_hat, (const half_t*)o_hat, M, n_active, (float*)scratch);
    k_style_loss_final<<<1, 1024, 0, s>>>((const float*)scratch, nb, M, n_active, StyleLossW{w_uniform, w_non_uniform, c_offset},
                                                  scale, fin);
    return lae::check_launch("style_loss_forward");
}

int lae_style_loss_backward(const void* w_logits, const void* o_raw, const float* palette, uint32_t P, uint32_t active_mask, uint32_t M,
                            const float* target, const float* fin, const float* upstream, float w_uniform, float w_non_uniform,
                            float c_offset, void* g_w_logits, void* g_o_raw, float* g_palette, void* scratch, void* stream) {
    if (!w_logits || !o_raw || !palette || !target || !fin || !upstream || !g_w_logits || !g_o_raw || !g_palette || !scratch) return LAE_ENULL;
    if (M == 0) return LAE_EINVAL;
    const int rc = check_palette(P, active_mask);
    if (rc) return rc;
    hipStream_t s = STREAM(stream);
    const uint32_t nb = lae::cdiv(M, PAL_BLOCK);
    const LossSrc ls{target, fin, upstream, StyleLossW{w_uniform, w_non_uniform, c_offset}};
    k_palette_bwd<true><<<nb, PAL_BLOCK, 0, s>>>((const half_t*)w_logits, (const half_t*)o_raw, palette, P, active_mask, M, nullptr, nullptr,
                                                 nullptr, (half_t*)g_w_logits, (half_t*)g_o_raw, (float*)scratch, ls);
    k_palette_grad_reduce<<<PAL_MAX * 3, 64, 0, s>>>((const float*)scratch, nb, P, active_mask, g_palette);
    return lae::check_launch("style_loss_backward");
}

}  // extern "C"
