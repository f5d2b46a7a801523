// ffmlp.hip -- fully fused bias-free fp16 MLP on the CDNA4 matrix cores.
//
// Replaces ffmlp/src/ffmlp.cu (+ the vendored CUTLASS split-K GEMMs) of the
// reference.  Layout contract (ffmlp.cu:631-634): weights = W0[hidden,in] |
// W1..[hidden,hidden] | Wout[16,hidden], every matrix [out,in] row-major, no
// bias; activations row-major [B, width] fp16; forward_buffer[l] holds the
// post-activation output of matmul l; backward_buffer[k] holds dL/d(output of
// matmul num_layers-1-k).
//
// MFMA mapping (v_mfma_f32_16x16x16_f16, wave64): the network is evaluated
// TRANSPOSED, H^T = W * X^T, i.e. the weight matrix is the A operand
// (M = out features) and the batch is the N dimension.  In that orientation the
// C/D fragment of M-tile t (lane = batch column, 4 regs = 4 consecutive features)
// is bit-for-bit the B fragment of K-step t of the next layer, so activations
// chain through all layers in registers: no LDS, no shuffles.  Accumulation is
// fp32 (the reference accumulates in fp16 fragments, ffmlp.cu:68).
#include <string>
#include <algorithm>
#include <type_traits>
#include <stdlib.h>
#include "lae_common.h"

namespace {

typedef _Float16 half_t;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma16(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

// K = 32 form (gfx950: the same 16 cycles as the K = 16 instruction, twice the work).  Lane group g owns k-slots
// 8g..8g+7 of BOTH operands; we fill slots 8g..8g+3 from k-step `lo` and 8g+4..8g+7 from k-step `hi`, i.e. the two
// 16-wide fragments a lane already holds in the K = 16 layout are simply concatenated.  A and B use the same
// permutation of k, so the contraction is unchanged and activations still chain layer to layer in registers.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f4 mfma32(h4 a_lo, h4 a_hi, h4 b_lo, h4 b_hi, f4 c) {
    const h8 a = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
    const h8 b = __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// acc += sum_{kt < KT} A(kt) * b[kt]: k-steps two at a time, one K = 16 step for an odd tail
template <int KT, typename LoadA>
__device__ __forceinline__ f4 mfma_ksteps(LoadA&& a_of, const h4* b, f4 acc) {
#pragma unroll
    for (int kt = 0; kt + 1 < KT; kt += 2) acc = mfma32(a_of(kt), a_of(kt + 1), b[kt], b[kt + 1], acc);
    // The odd tail is a K = 32 step with zeros in the upper k-slots, NOT the K = 16 instruction: a v_mfma_f32_16x16x16_f16
    // whose accumulator input is the result of the v_mfma_f32_16x16x32_f16 issued just before it read a stale accumulator in
    // k_mlp_bwd_wave<48, 1> (second tile's first-layer sums wrong, correct again with this form or with unrelated code moved:
    // tests/test_gpu_ffmlp.py[*-48-64-2], tools/mlp_debug.py) -- the compiler's wait-state padding between the two shapes is
    // not enough on gfx950.  Same cycles (the K = 16 form occupies the pipe as long as the K = 32 form), same sums.
    if constexpr (KT & 1) {
        const h4 z = h4{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
        acc = mfma32(a_of(KT - 1), z, b[KT - 1], z, acc);
    }
    return acc;
}

#define K_ACT 10.0f   // ffmlp/src/utils.h:41
__device__ __forceinline__ float act_fwd(uint32_t a, float v) {           // utils.h:424-470
    switch (a) {
        case LAE_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case LAE_ACT_EXPONENTIAL: return expf(v);
        case LAE_ACT_SINE: return sinf(v);
        case LAE_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case LAE_ACT_SQUAREPLUS: { const float x = v * K_ACT; return 0.5f * (x + sqrtf(x * x + 4)) / K_ACT; }
        case LAE_ACT_SOFTPLUS: return logf(expf(v * K_ACT) + 1.0f) / K_ACT;
        default: return v;
    }
}
__device__ __forceinline__ float act_bwd(uint32_t a, float g, float fwd) { // utils.h:537-582
    switch (a) {
        case LAE_ACT_RELU: return fwd > 0.0f ? g : 0.0f;
        case LAE_ACT_EXPONENTIAL: return g * fwd;
        case LAE_ACT_SINE: return g;
        case LAE_ACT_SIGMOID: return g * (fwd * (1 - fwd));
        case LAE_ACT_SQUAREPLUS: { const float y = fwd * K_ACT; return g * (y * y / (y * y + 1)); }
        case LAE_ACT_SOFTPLUS: return g * (1.0f - expf(-fwd * K_ACT));
        default: return g;
    }
}

constexpr int MLP_BLOCK = 256;   // 4 waves

// store a D-layout tile set (features 16*mt + 4g + r of batch row `row`) as row-major halves
template <int MT>
__device__ __forceinline__ void store_tiles(half_t* __restrict__ dst, size_t row, uint32_t width, int g, const h4 (&h)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++) *reinterpret_cast<h4*>(dst + row * width + mt * 16 + 4 * g) = h[mt];
}

// ---------------------------------------------------------------- forward / inference
// ffmlp.cu:331-407 (kernel_mlp_fused).  One wave owns NT tiles of 16 batch rows at a time.
template <int WIDTH, int NT>
__global__ __launch_bounds__(MLP_BLOCK) void k_mlp_fwd(
    const half_t* __restrict__ in, const half_t* __restrict__ W, uint32_t in_dim, uint32_t n_hidden, uint32_t act,
    uint32_t out_act, half_t* __restrict__ fwd_buf, half_t* __restrict__ out, uint32_t B, uint32_t n_groups) {
    constexpr int MT = WIDTH / 16;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const uint32_t wave0 = blockIdx.x * (MLP_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = gridDim.x * (MLP_BLOCK / 64);
    for (uint32_t grp = wave0; grp < n_groups; grp += nwaves) {
        const size_t row0 = (size_t)grp * 16 * NT;
        f4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f4{0, 0, 0, 0};
        // input layer, K = in_dim
        for (uint32_t kt = 0; kt < in_dim / 16; kt++) {
            h4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
                b[nt] = *reinterpret_cast<const h4*>(in + (row0 + nt * 16 + c) * in_dim + kt * 16 + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const h4 a = *reinterpret_cast<const h4*>(W + (size_t)(mt * 16 + c) * in_dim + kt * 16 + 4 * g);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16(a, b[nt], acc[mt][nt]);
            }
        }
        h4 h[NT][MT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) h[nt][mt][r] = (half_t)act_fwd(act, acc[mt][nt][r]);
            if (fwd_buf) store_tiles<MT>(fwd_buf, row0 + nt * 16 + c, WIDTH, g, h[nt]);
        }
        const half_t* Wl = W + (size_t)WIDTH * in_dim;
        for (uint32_t l = 0; l < n_hidden; l++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f4{0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < MT; kt++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const h4 a = *reinterpret_cast<const h4*>(Wl + (size_t)(mt * 16 + c) * WIDTH + kt * 16 + 4 * g);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16(a, h[nt][kt], acc[mt][nt]);
                }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) h[nt][mt][r] = (half_t)act_fwd(act, acc[mt][nt][r]);
                if (fwd_buf) store_tiles<MT>(fwd_buf + (size_t)(l + 1) * B * WIDTH, row0 + nt * 16 + c, WIDTH, g, h[nt]);
            }
            Wl += (size_t)WIDTH * WIDTH;
        }
        // output layer: 16 padded outputs (ffmlp.cu:242-302)
        f4 o[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) o[nt] = f4{0, 0, 0, 0};
#pragma unroll
        for (int kt = 0; kt < MT; kt++) {
            const h4 a = *reinterpret_cast<const h4*>(Wl + (size_t)c * WIDTH + kt * 16 + 4 * g);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) o[nt] = mfma16(a, h[nt][kt], o[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            h4 v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = (half_t)act_fwd(out_act, o[nt][r]);
            *reinterpret_cast<h4*>(out + (row0 + nt * 16 + c) * 16 + 4 * g) = v;
        }
    }
}

// transposed A fragment: A[row = c][k = 4g + j] = Wm[(k0 + 4g + j) * ld + col0 + c]
__device__ __forceinline__ h4 load_wT(const half_t* __restrict__ Wm, uint32_t ld, uint32_t k0, uint32_t col0, int c, int g) {
    const half_t* p = Wm + (size_t)(k0 + 4 * g) * ld + col0 + c;
    h4 a;
    a[0] = p[0]; a[1] = p[ld]; a[2] = p[2 * (size_t)ld]; a[3] = p[3 * (size_t)ld];
    return a;
}

// ---------------------------------------------------------------- backward through the activations
// ffmlp.cu:410-518 (kernel_mlp_fused_backward) + the dL/dinput GEMM of :880-887
template <int WIDTH, int NT>
__global__ __launch_bounds__(MLP_BLOCK) void k_mlp_bwd(
    const half_t* __restrict__ grad, const half_t* __restrict__ W, const half_t* __restrict__ fwd_buf, uint32_t in_dim,
    uint32_t n_hidden, uint32_t act, half_t* __restrict__ bwd_buf, half_t* __restrict__ grad_in, uint32_t B,
    uint32_t n_groups) {
    constexpr int MT = WIDTH / 16;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const uint32_t wave0 = blockIdx.x * (MLP_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = gridDim.x * (MLP_BLOCK / 64);
    const half_t* Wout = W + (size_t)WIDTH * in_dim + (size_t)WIDTH * WIDTH * n_hidden;
    for (uint32_t grp = wave0; grp < n_groups; grp += nwaves) {
        const size_t row0 = (size_t)grp * 16 * NT;
        f4 acc[MT][NT];
        h4 d[NT][MT];
        // output layer: dH = Wout^T dY, K = 16 outputs
        {
            h4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b[nt] = *reinterpret_cast<const h4*>(grad + (row0 + nt * 16 + c) * 16 + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const h4 a = load_wT(Wout, WIDTH, 0, mt * 16, c, g);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16(a, b[nt], f4{0, 0, 0, 0});
            }
            const half_t* f = fwd_buf + (size_t)n_hidden * B * WIDTH;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const h4 fv = *reinterpret_cast<const h4*>(f + (row0 + nt * 16 + c) * WIDTH + mt * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; r++) d[nt][mt][r] = (half_t)act_bwd(act, acc[mt][nt][r], (float)fv[r]);
                }
                store_tiles<MT>(bwd_buf, row0 + nt * 16 + c, WIDTH, g, d[nt]);
            }
        }
        // hidden layers, last to first
        for (uint32_t k = 0; k < n_hidden; k++) {
            const half_t* Wl = W + (size_t)WIDTH * in_dim + (size_t)WIDTH * WIDTH * (n_hidden - 1 - k);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f4{0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < MT; kt++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const h4 a = load_wT(Wl, WIDTH, kt * 16, mt * 16, c, g);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16(a, d[nt][kt], acc[mt][nt]);
                }
            const half_t* f = fwd_buf + (size_t)(n_hidden - 1 - k) * B * WIDTH;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const h4 fv = *reinterpret_cast<const h4*>(f + (row0 + nt * 16 + c) * WIDTH + mt * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; r++) d[nt][mt][r] = (half_t)act_bwd(act, acc[mt][nt][r], (float)fv[r]);
                }
                store_tiles<MT>(bwd_buf + (size_t)(k + 1) * B * WIDTH, row0 + nt * 16 + c, WIDTH, g, d[nt]);
            }
        }
        // dL/dinput = W0^T dH0 (no activation)
        if (grad_in) {
            for (uint32_t it = 0; it < in_dim / 16; it++) {
                f4 gi[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) gi[nt] = f4{0, 0, 0, 0};
#pragma unroll
                for (int kt = 0; kt < MT; kt++) {
                    const h4 a = load_wT(W, in_dim, kt * 16, it * 16, c, g);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) gi[nt] = mfma16(a, d[nt][kt], gi[nt]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    h4 v;
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = (half_t)gi[nt][r];
                    *reinterpret_cast<h4*>(grad_in + (row0 + nt * 16 + c) * in_dim + it * 16 + 4 * g) = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradients
// dW[o][i] = sum_b dY[b][o] * A[b][i]  (ffmlp.cu:800-877, the reference's split-K CUTLASS GEMMs).
// blockIdx.y = job = (matrix, 64x64 output block); blockIdx.x = batch slice.  Each wave
// accumulates a 4x4 block of 16x16 tiles over its rows, the 4 waves of the group are summed in
// LDS, and one fp32 slab per (slice, matrix) goes to the workspace; k_dw_reduce sums the slabs
// in slice order (deterministic) and rounds once to fp16.
struct DwJob { uint32_t mat, mb, nb; };

__device__ __forceinline__ DwJob decode_job(uint32_t job, uint32_t MB, uint32_t NB0, uint32_t n_hidden) {
    DwJob j;
    if (job < MB * NB0) { j.mat = 0; j.mb = job / NB0; j.nb = job % NB0; return j; }
    job -= MB * NB0;
    const uint32_t per = MB * MB;
    if (job < per * n_hidden) { j.mat = 1 + job / per; const uint32_t r = job % per; j.mb = r / MB; j.nb = r % MB; return j; }
    job -= per * n_hidden;
    j.mat = n_hidden + 1; j.mb = 0; j.nb = job;
    return j;
}

template <int WIDTH>
__global__ __launch_bounds__(MLP_BLOCK) void k_mlp_dw(
    const half_t* __restrict__ grad, const half_t* __restrict__ inputs, const half_t* __restrict__ fwd_buf,
    const half_t* __restrict__ bwd_buf, uint32_t B, uint32_t in_dim, uint32_t n_hidden, float* __restrict__ slabs,
    uint32_t nW, uint32_t rows_per_slice) {
    constexpr uint32_t MT = WIDTH / 16;
    constexpr uint32_t MB = (MT + 3) / 4;
    const uint32_t NB0 = (in_dim / 16 + 3) / 4;
    const DwJob job = decode_job(blockIdx.y, MB, NB0, n_hidden);
    // operands of this matrix
    const half_t* dY; const half_t* A; uint32_t OUT, IN; size_t w_off;
    if (job.mat == 0) { dY = bwd_buf + (size_t)n_hidden * B * WIDTH; A = inputs; OUT = WIDTH; IN = in_dim; w_off = 0; }
    else if (job.mat <= n_hidden) {
        dY = bwd_buf + (size_t)(n_hidden - job.mat) * B * WIDTH; A = fwd_buf + (size_t)(job.mat - 1) * B * WIDTH;
        OUT = WIDTH; IN = WIDTH; w_off = (size_t)WIDTH * in_dim + (size_t)WIDTH * WIDTH * (job.mat - 1);
    } else { dY = grad; A = fwd_buf + (size_t)n_hidden * B * WIDTH; OUT = 16; IN = WIDTH;
             w_off = (size_t)WIDTH * in_dim + (size_t)WIDTH * WIDTH * n_hidden; }
    const uint32_t mt0 = job.mb * 4, nt0 = job.nb * 4;
    const uint32_t mt_n = min(4u, OUT / 16 - mt0), nt_n = min(4u, IN / 16 - nt0);

    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // scalar: per-wave branches stay uniform
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f4{0, 0, 0, 0};

    const size_t r_begin = (size_t)blockIdx.x * rows_per_slice;
    const size_t r_end = min((size_t)B, r_begin + rows_per_slice);
    for (size_t b0 = r_begin + (size_t)w * 16; b0 < r_end; b0 += 16 * (MLP_BLOCK / 64)) {
        h4 a[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if ((uint32_t)i < mt_n) {
                const half_t* p = dY + (b0 + 4 * g) * OUT + (mt0 + i) * 16 + c;
                a[i][0] = p[0]; a[i][1] = p[OUT]; a[i][2] = p[2 * (size_t)OUT]; a[i][3] = p[3 * (size_t)OUT];
            }
            if ((uint32_t)i < nt_n) {
                const half_t* q = A + (b0 + 4 * g) * IN + (nt0 + i) * 16 + c;
                bf[i][0] = q[0]; bf[i][1] = q[IN]; bf[i][2] = q[2 * (size_t)IN]; bf[i][3] = q[3 * (size_t)IN];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((uint32_t)i < mt_n && (uint32_t)j < nt_n) acc[i][j] = mfma16(a[i], bf[j], acc[i][j]);
    }
    // sum the 4 waves in LDS (wave order fixed by the barriers -> deterministic)
    __shared__ float red[16 * 256];
    for (int ww = 0; ww < MLP_BLOCK / 64; ww++) {
        if (w == ww) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float* p = &red[(i * 4 + j) * 256 + (4 * g + r) * 16 + c];
                        *p = (ww == 0) ? acc[i][j][r] : (*p + acc[i][j][r]);
                    }
        }
        __syncthreads();
    }
    float* slab = slabs + (size_t)blockIdx.x * nW + w_off;
    for (uint32_t e = threadIdx.x; e < 16 * 256; e += MLP_BLOCK) {
        const uint32_t t = e >> 8, i = t >> 2, j = t & 3, rr = (e >> 4) & 15, cc = e & 15;
        if (i < mt_n && j < nt_n) slab[(size_t)((mt0 + i) * 16 + rr) * IN + (nt0 + j) * 16 + cc] = red[e];
    }
}

// sum the per-slice slabs in a fixed order (deterministic) and round once to fp16.
// 1024 threads = 64 weights x 16 slice groups, 8 slab loads in flight per lane (the kernel is pure load latency: 512
// slabs of 45 KB); the 16 partial sums are combined in a fixed tree.
// accumulate != 0: gw += sum (the optimizer's persistent gradient buffer) instead of gw = sum.
constexpr int DWR_GROUPS = 16;
__device__ __forceinline__ void dw_reduce_body(const float* __restrict__ slabs, uint32_t n_slices, uint32_t nW, half_t* __restrict__ gw,
                                               int accumulate, uint32_t block, int32_t* __restrict__ nf_flag = nullptr) {
    __shared__ float part[DWR_GROUPS][64];
    const uint32_t e = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const uint32_t i = block * 64 + e;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < nW) {
        uint32_t k = sg;
        for (; k + 7 * DWR_GROUPS < n_slices; k += 8 * DWR_GROUPS) {
#pragma unroll
            for (int u = 0; u < 8; u++) s[u] += slabs[(size_t)(k + u * DWR_GROUPS) * nW + i];
        }
        for (; k < n_slices; k += DWR_GROUPS) s[0] += slabs[(size_t)k * nW + i];
    }
    part[sg][e] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (sg == 0 && i < nW) {
        float t[DWR_GROUPS];
#pragma unroll
        for (int u = 0; u < DWR_GROUPS; u++) t[u] = part[u][e];
#pragma unroll
        for (int w = DWR_GROUPS / 2; w > 0; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; u++) t[u] = t[u] + t[u + w];
        const half_t r = accumulate ? (half_t)((float)gw[i] + t[0]) : (half_t)t[0];
        gw[i] = r;
        // the caller's flag (the optimizer's found_inf word): a non-finite weight gradient is reported where it is stored
        if (nf_flag && (__builtin_bit_cast(uint16_t, r) & 0x7c00u) == 0x7c00u) atomicOr(nf_flag, 1);
    }
}
__global__ __launch_bounds__(64 * DWR_GROUPS) void k_dw_reduce(const float* __restrict__ slabs, uint32_t n_slices, uint32_t nW,
                                                               half_t* __restrict__ gw, int accumulate = 0, int32_t* __restrict__ nf_flag = nullptr) {
    dw_reduce_body(slabs, n_slices, nW, gw, accumulate, blockIdx.x, nf_flag);
}
// the same reduction for two networks in one launch (blocks [0, nb_a) reduce A, the rest B): one graph node less per step
// One block more than the reduction needs takes the deferred loss value along (lae_composite_rays_train_step with
// defer_loss): the fixed-order sum of the criterion's per-workgroup partials -> loss_out[0] = mean * scale, [1] = mean,
// the arithmetic of k_loss_finish (raymarching.hip), without a launch of its own on the step's critical path.
struct LossFinish { const float* partials; uint32_t n_part, n_elem; const float* scale; float* out; };
__global__ __launch_bounds__(64 * DWR_GROUPS) void k_dw_reduce2(const float* __restrict__ slabs_a, uint32_t n_a, uint32_t nW_a,
                                                                half_t* __restrict__ gw_a, const float* __restrict__ slabs_b,
                                                                uint32_t n_b, uint32_t nW_b, half_t* __restrict__ gw_b,
                                                                uint32_t nb_a, uint32_t nb_b, int accumulate, int32_t* __restrict__ nf_flag,
                                                                LossFinish lf) {
    if (blockIdx.x < nb_a) dw_reduce_body(slabs_a, n_a, nW_a, gw_a, accumulate, blockIdx.x, nf_flag);
    else if (blockIdx.x < nb_a + nb_b) dw_reduce_body(slabs_b, n_b, nW_b, gw_b, accumulate, blockIdx.x - nb_a, nf_flag);
    else {
        static_assert(64 * DWR_GROUPS == 1024, "the loss sum is written for 16 waves, like k_loss_finish");
        __shared__ float part[16];
        float acc = 0.0f;
        for (uint32_t i = threadIdx.x; i < lf.n_part; i += 1024) acc += lf.partials[i];
        {   // the wave sum of raymarching.hip (wave_sum = last lane of the DPP inclusive scan), statement for statement: the
            // deferred value has the same bits as the one k_loss_finish writes
            auto dpp = [](float v, auto ctrl, auto mask) {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, decltype(mask)::value, 0xf, false));
            };
            using std::integral_constant;
            acc += dpp(acc, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});
            acc += dpp(acc, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});
            acc += dpp(acc, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});
            acc += dpp(acc, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});
            acc += dpp(acc, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
            acc += dpp(acc, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
            acc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 63));
        }
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 16; w++) t += part[w];
            const float loss = t / (float)lf.n_elem;
            lf.out[0] = loss * (lf.scale ? lf.scale[0] : 1.0f);
            lf.out[1] = loss;
        }
    }
}

// ---------------------------------------------------------------- fused NeRF head (network_ff.py:51-81 in one kernel)
// sigma net (32 -> 64 -> 64 -> 16) -> sigma = density_scale * exp(h[0]) ; colour-net input = [SH_4(dir) | h[1..15] | 0]
// -> colour net (32 -> 64 -> 64 -> 64 -> 16) -> rgb = sigmoid(out[0..2]).  One wave owns a 16-row tile and chains
// every activation through registers in the MFMA C/D layout (lane = row, 4 regs = 4 consecutive features):
//   * h[1 + m] -> colour feature 16 + m is a shift by one feature: three in-lane register moves and one
//     16-lane shuffle;
//   * the 16 SH values of a row are evaluated per lane and the lane keeps components 4g..4g+3 (its K-slice);
//   * weights of both nets sit row-major in LDS (ds_read_b64 fragments).
// Saved for the backward: h [M,16] fp16 (32 B/row) and rgb; nothing else touches HBM.
#include "sh_table.inc"

__device__ __forceinline__ void stage_rows(half_t* dst, int ld, const half_t* __restrict__ src, int rows, int cols) {
    for (uint32_t e = threadIdx.x; e < (uint32_t)rows * cols / 4; e += blockDim.x) {
        const uint32_t r = (e * 4) / cols, k = (e * 4) % cols;
        *reinterpret_cast<h4*>(dst + r * ld + k) = *reinterpret_cast<const h4*>(src + (size_t)r * cols + k);
    }
}

// one 64-wide layer on a 16-row tile: acc[mt] = sum_kt W[mt*16+c][kt*16+4g..] * in[kt]
template <int KT>
__device__ __forceinline__ void layer64(const half_t* Wl, int ld, const h4 (&in)[KT], int c, int g, f4 (&acc)[4]) {
    // k-steps outside, the four output tiles inside: four independent accumulators in flight, so consecutive MFMAs never
    // wait for one another's result (tile by tile the compiler reused one accumulator and padded with s_nop)
#pragma unroll
    for (int mt = 0; mt < 4; mt++) acc[mt] = f4{0, 0, 0, 0};
#pragma unroll
    for (int kt = 0; kt + 1 < KT; kt += 2)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const half_t* wr = Wl + (mt * 16 + c) * ld + kt * 16 + 4 * g;
            acc[mt] = mfma32(*reinterpret_cast<const h4*>(wr), *reinterpret_cast<const h4*>(wr + 16), in[kt], in[kt + 1], acc[mt]);
        }
    if constexpr (KT & 1) {                                  // odd tail: a K = 32 step with zero upper k-slots (see mfma_ksteps)
        const h4 z = h4{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
            acc[mt] = mfma32(*reinterpret_cast<const h4*>(Wl + (mt * 16 + c) * ld + (KT - 1) * 16 + 4 * g), z, in[KT - 1], z, acc[mt]);
    }
}
// ReLU + fp16 rounding of a 64-wide layer.  round(max(x, 0)) == max(round(x), 0), so the maximum is taken on the packed
// halves (v_cvt_pk_f16_f32 + v_pk_max_f16: 2 values per instruction; fmaxf on the floats costs a canonicalising
// v_max_f32 plus the real one per value -- 160 of this kernel's ~700 VALU instructions per tile)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void relu4(const f4 (&acc)[4], h4 (&out)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        h2 lo = {(half_t)acc[mt][0], (half_t)acc[mt][1]}, hi = {(half_t)acc[mt][2], (half_t)acc[mt][3]};
        lo = __builtin_elementwise_max(lo, h2{(half_t)0.0f, (half_t)0.0f});
        hi = __builtin_elementwise_max(hi, h2{(half_t)0.0f, (half_t)0.0f});
        out[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
    }
}
__device__ __forceinline__ f4 out16(const half_t* Wl, int ld, const h4 (&in)[4], int c, int g) {
    return mfma_ksteps<4>([&](int kt) { return *reinterpret_cast<const h4*>(Wl + c * ld + kt * 16 + 4 * g); }, in, f4{0, 0, 0, 0});
}

// colour-net input fragments of a tile: k-step 0 = SH components 4g..4g+3 of the row's direction,
// k-step 1 = features 16 + 4g + j = h[1 + 4g + j] (h = fp16 sigma-net output in C/D layout), feature 31 = 0
__device__ __forceinline__ void color_inputs(float dx, float dy, float dz, const h4& hq, int g, h4 (&cin)[2]);
__device__ __forceinline__ void color_inputs(const float* __restrict__ dirs, size_t row, const h4& hq, int g, h4 (&cin)[2]) {
    color_inputs(dirs[3 * row], dirs[3 * row + 1], dirs[3 * row + 2], hq, g, cin);
}
__device__ __forceinline__ void color_inputs(float dx, float dy, float dz, const h4& hq, int g, h4 (&cin)[2]) {
    float o[16], gx[1], gy[1], gz[1];
    sh_eval<4, false>(dx, dy, dz, o, gx, gy, gz);
    // the lane keeps components 4g .. 4g + 3.  Every lane evaluates all 16 (~40 instructions) and picks with bit masks:
    // written as `g == 0 ? o[j] : g == 1 ? ...` the compiler sinks the polynomials into four divergent branches, which a
    // wave (all four g groups) then runs one after the other -- ~190 of the ~600 instructions per 16-row tile
    const uint32_t m0 = g == 0 ? 0xffffffffu : 0u, m1 = g == 1 ? 0xffffffffu : 0u, m2 = g == 2 ? 0xffffffffu : 0u,
                   m3 = g == 3 ? 0xffffffffu : 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t bits = (__builtin_bit_cast(uint32_t, o[j]) & m0) | (__builtin_bit_cast(uint32_t, o[4 + j]) & m1) |
                              (__builtin_bit_cast(uint32_t, o[8 + j]) & m2) | (__builtin_bit_cast(uint32_t, o[12 + j]) & m3);
        cin[0][j] = (half_t)__builtin_bit_cast(float, bits);
    }
    const half_t nxt = __builtin_bit_cast(half_t, (uint16_t)__shfl_down((int)__builtin_bit_cast(uint16_t, hq[0]), 16, 64));
    cin[1][0] = hq[1]; cin[1][1] = hq[2]; cin[1][2] = hq[3];
    cin[1][3] = g == 3 ? (half_t)0.0f : nxt;
}

// ---------------------------------------------------------------- fused backward (WIDTH = 64, ReLU)
// One kernel for everything the reference does in kernel_mlp_fused_backward + (num_layers+1) split-K CUTLASS
// GEMMs (ffmlp.cu:410-518, 800-877), WITHOUT the forward/backward buffers: the hidden activations are
// recomputed from the 64-byte input row (cheap on the matrix cores; the buffers cost ~3.5 KB/sample of HBM
// traffic), dL/dH is chained through the layers in registers exactly like k_mlp_bwd, and every wave accumulates
// its own dW = dY^T A for all weight tiles over its 16-row tiles:
//   * the dY / A tiles are written row-major into a WAVE-PRIVATE LDS region with ds_write_b64 straight from the
//     MFMA C/D layout and read back transposed with ds_read_b64_tr_b16 -- that IS the A / B operand layout for a
//     product that sums over the batch (lane) index; no barrier, waves never share tiles;
//   * weights live row-major in LDS (padded rows): forward fragments are plain ds_read_b64, the transposed
//     fragments of the backward chain are ds_read_b64_tr_b16 of the same image;
//   * the 4 waves' dW accumulators are summed through LDS once at the end (fixed order) and stored as one fp32
//     slab per workgroup; k_dw_reduce adds the slabs in order -> deterministic gradients.
typedef __fp16 hraw4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h4 lds_tr_read(const half_t* p) {
    const hraw4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hraw4*)(p));
    return __builtin_bit_cast(h4, r);
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// B fragments of a 32-wide encoder row.  Row-major [M,32]: features 16kt + 4g .. +3 are one 8-byte load.  Level-major
// [16][M][2] (the grid kernels' native layout): the same four features are the channel pairs of levels 8kt + 2g and
// 8kt + 2g + 1 -- two 4-byte loads, 64 contiguous bytes per level for the 16 rows of a tile.
__device__ __forceinline__ void load_enc_frags(const half_t* __restrict__ enc, size_t row, size_t M, int g, int level_major, h4 (&xf)[2]) {
#pragma unroll
    for (int kt = 0; kt < 2; kt++) {
        if (level_major) {
            const uint32_t lo = *reinterpret_cast<const uint32_t*>(enc + ((size_t)(8 * kt + 2 * g) * M + row) * 2);
            const uint32_t hi = *reinterpret_cast<const uint32_t*>(enc + ((size_t)(8 * kt + 2 * g + 1) * M + row) * 2);
            const uint2 v{lo, hi};
            xf[kt] = __builtin_bit_cast(h4, v);
        } else {
            xf[kt] = *reinterpret_cast<const h4*>(enc + row * 32 + kt * 16 + 4 * g);
        }
    }
}

// MODE 1 = colour net of the fused NeRF head: the input rows are built on the fly from (h, dirs) exactly like
// k_nerf_head_fwd, dL/dout comes from (grad_rgb, rgb) through the sigmoid, and instead of dL/dX the kernel writes
// dL/dh [M,16] = [ grad_sigma * density_scale * exp(clamp(h0, -15, 15)) | dX[16..30] ] (trunc_exp backward,
// activation.py:14-17, and the inverse of the one-feature shift) -- the sigma net's backward reads that directly.
struct HeadBwdArgs {
    const float* dirs; const float* rgbs; const float* grad_rgbs; const float* grad_sigmas; float density_scale;
    int level_major;        // MODE 0, IN = 32: x and grad_in are [16][B][2] (the grid kernels' layout) instead of [B,32]
};

// ---------------------------------------------------------------- fused backward, workgroup-cooperative dW
// Same mathematics as k_mlp_bwd_fused, different ownership of the weight-gradient tiles.  There every wave accumulates
// ALL dW tiles of the net over its own rows (176 accumulator registers for the colour net -> 2 waves/SIMD with spills,
// and a cross-wave reduction at the end).  Here a workgroup of WAVES waves shares one [16*WAVES rows] tile of X, H, D, G
// in LDS: each wave still runs the forward recompute and the dL/dH chain for its own 16 rows in registers, but after a
// barrier the dW tiles are DIVIDED among the waves, each contracted over all 16*WAVES rows with K = 32 MFMAs.  Per wave
// that is 6 accumulator tiles instead of 44, half as many dW MFMAs, no final reduction, and 4 waves/SIMD.
template <int IN, int NH, int WAVES>
struct CoopCfg {
    static constexpr int KT0 = IN / 16;
    static constexpr int R = 16 * WAVES;                   // rows per workgroup iteration
    static constexpr int LDX = IN + 8, LDH = 72, LDG = 24;
    static constexpr int W0_OFF = 0;
    static constexpr int WH_OFF = 64 * LDX;
    static constexpr int WO_OFF = WH_OFF + NH * 64 * LDH;
    static constexpr int W_HALVES = WO_OFF + 16 * LDH;
    static constexpr int X_OFF = W_HALVES;
    static constexpr int H_OFF = X_OFF + R * LDX;
    static constexpr int D_OFF = H_OFF + R * LDH;
    static constexpr int G_OFF = D_OFF + R * LDH;
    static constexpr int LDS_HALVES = G_OFF + R * LDG;
    static constexpr int TH = 16 / WAVES > 0 ? 16 / WAVES : 1;                      // hidden-layer dW tiles per wave
    static constexpr int T0 = (4 * KT0 + WAVES - 1) / WAVES;                        // input-layer dW tiles per wave
};

template <int IN, int NH, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_mlp_bwd_coop(
    const half_t* __restrict__ grad, const half_t* __restrict__ x, const half_t* __restrict__ W, uint32_t n_tiles,
    half_t* __restrict__ grad_in, float* __restrict__ slabs, uint32_t nW, HeadBwdArgs ha) {
    using C = CoopCfg<IN, NH, WAVES>;
    constexpr int KT0 = C::KT0, R = C::R, NT = 64 * WAVES;
    static_assert(16 % WAVES == 0 && WAVES >= 4, "dW tiles are dealt round-robin to 4, 8 or 16 waves");
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* Wl = lds;
    half_t* Xs = lds + C::X_OFF; half_t* Hs = lds + C::H_OFF; half_t* Ds = lds + C::D_OFF; half_t* Gs = lds + C::G_OFF;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    // wave index as a SCALAR: the per-wave branches below (`w < 4`, `t < 4 * KT0`) must be uniform branches.  As a vector
    // value they compile to EXEC-masked regions; with IN = 48 (the only shape where waves 4-7 skip a dW0 tile) one 16-row
    // tile of dX / dW came out wrong in ~25 % of the launches (tests/test_gpu_ffmlp.py [1152-48-64-2]).
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int r0 = 16 * w;                                   // this wave's rows inside the shared tiles

    for (uint32_t e = threadIdx.x; e < 64u * IN / 4; e += NT) {
        const uint32_t r = (e * 4) / IN, k = (e * 4) % IN;
        *reinterpret_cast<h4*>(Wl + C::W0_OFF + r * C::LDX + k) = *reinterpret_cast<const h4*>(W + (size_t)r * IN + k);
    }
    for (uint32_t e = threadIdx.x; e < (uint32_t)NH * 64 * 64 / 4; e += NT) {
        const uint32_t m = (e * 4) / 4096, r = ((e * 4) % 4096) / 64, k = (e * 4) % 64;
        *reinterpret_cast<h4*>(Wl + C::WH_OFF + (m * 64 + r) * C::LDH + k) =
            *reinterpret_cast<const h4*>(W + 64 * IN + (size_t)m * 4096 + r * 64 + k);
    }
    for (uint32_t e = threadIdx.x; e < 16u * 64 / 4; e += NT) {
        const uint32_t r = (e * 4) / 64, k = (e * 4) % 64;
        *reinterpret_cast<h4*>(Wl + C::WO_OFF + r * C::LDH + k) =
            *reinterpret_cast<const h4*>(W + 64 * IN + (size_t)NH * 4096 + r * 64 + k);
    }

    f4 aO = f4{0, 0, 0, 0};
    f4 aH[NH][C::TH], a0[C::T0];
#pragma unroll
    for (int m = 0; m < NH; m++)
#pragma unroll
        for (int i = 0; i < C::TH; i++) aH[m][i] = f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < C::T0; i++) a0[i] = f4{0, 0, 0, 0};

    // dW tile (mt, nt) += A^T B over the R shared rows: A block mt of tile `At` (ld lda), B block nt of tile `Bt` (ld ldb)
    auto dw_tile = [&](const half_t* At, int lda, int mt, const half_t* Bt, int ldb, int nt, f4 acc) {
#pragma unroll
        for (int ks = 0; ks < R / 32; ks++) {
            const half_t* ar = At + (32 * ks + 4 * g + tq) * lda + mt * 16 + 4 * tp;
            const half_t* br = Bt + (32 * ks + 4 * g + tq) * ldb + nt * 16 + 4 * tp;
            acc = mfma32(lds_tr_read(ar), lds_tr_read(ar + 16 * lda), lds_tr_read(br), lds_tr_read(br + 16 * ldb), acc);
        }
        return acc;
    };

    const uint32_t n_groups = (n_tiles + WAVES - 1) / WAVES;
    for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint32_t tile = grp * WAVES + w;
        const bool active = tile < n_tiles;
        const size_t row = (size_t)(active ? tile : 0) * 16 + c;
        const h4 zero4 = h4{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
        // ---- inputs of this wave's 16 rows
        h4 xf[KT0];
        h4 gf = zero4, hq = zero4;
        if constexpr (MODE == 1) {
            hq = *reinterpret_cast<const h4*>(x + row * 16 + 4 * g);
            color_inputs(ha.dirs, row, hq, g, xf);
            if (g == 0) {
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const float y = ha.rgbs[row * 3 + r];
                    gf[r] = (half_t)(ha.grad_rgbs[row * 3 + r] * y * (1.0f - y));
                }
            }
        } else {
            if constexpr (KT0 == 2) {
                load_enc_frags(x, row, (size_t)n_tiles * 16, g, ha.level_major, xf);
            } else {
#pragma unroll
                for (int kt = 0; kt < KT0; kt++) xf[kt] = *reinterpret_cast<const h4*>(x + row * IN + kt * 16 + 4 * g);
            }
            gf = *reinterpret_cast<const h4*>(grad + row * 16 + 4 * g);
        }
        if (!active) {                                       // rows past the batch contribute nothing
#pragma unroll
            for (int kt = 0; kt < KT0; kt++) xf[kt] = zero4;
            gf = zero4;
        }
        __syncthreads();                                     // S0: readers of X / G / H / D of the previous group are done
#pragma unroll
        for (int kt = 0; kt < KT0; kt++) *reinterpret_cast<h4*>(Xs + (r0 + c) * C::LDX + kt * 16 + 4 * g) = xf[kt];
        *reinterpret_cast<h4*>(Gs + (r0 + c) * C::LDG + 4 * g) = gf;
        // ---- recompute the hidden activations of the own rows (registers)
        h4 h[NH + 1][4];
        {
            f4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
                acc[mt] = mfma_ksteps<KT0>([&](int kt) { return *reinterpret_cast<const h4*>(Wl + C::W0_OFF + (mt * 16 + c) * C::LDX + kt * 16 + 4 * g); },
                                           xf, f4{0, 0, 0, 0});
            relu4(acc, h[0]);
#pragma unroll
            for (int l = 0; l < NH; l++) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
                    acc[mt] = mfma_ksteps<4>([&](int kt) { return *reinterpret_cast<const h4*>(Wl + C::WH_OFF + (l * 64 + mt * 16 + c) * C::LDH + kt * 16 + 4 * g); },
                                             h[l], f4{0, 0, 0, 0});
                relu4(acc, h[l + 1]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; mt++) *reinterpret_cast<h4*>(Hs + (r0 + c) * C::LDH + mt * 16 + 4 * g) = h[NH][mt];
        __syncthreads();                                     // S1: G and H_NH of all rows are in LDS
        // ---- output layer: dWout tile nt = w (waves 0..3) ; own rows: dH_NH = (Wout^T G) * relu'
        if (w < 4) aO = dw_tile(Gs, C::LDG, 0, Hs, C::LDH, w, aO);
        h4 d[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const h4 a = lds_tr_read(Wl + C::WO_OFF + (4 * g + tq) * C::LDH + mt * 16 + 4 * tp);             // A[f][o] = Wout[o][f]
            const f4 acc = mfma16(a, gf, f4{0, 0, 0, 0});
#pragma unroll
            for (int r = 0; r < 4; r++) d[mt][r] = ((float)h[NH][mt][r] > 0.0f) ? (half_t)acc[r] : (half_t)0.0f;
        }
        // ---- hidden layers, last to first
#pragma unroll
        for (int l = NH; l >= 1; l--) {
            __syncthreads();                                 // S2: readers of H and D are done
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                *reinterpret_cast<h4*>(Ds + (r0 + c) * C::LDH + mt * 16 + 4 * g) = d[mt];
                *reinterpret_cast<h4*>(Hs + (r0 + c) * C::LDH + mt * 16 + 4 * g) = h[l - 1][mt];
            }
            __syncthreads();                                 // S3: D_l and H_{l-1} of all rows are in LDS
#pragma unroll
            for (int i = 0; i < C::TH; i++) {
                const int t = w + i * WAVES;                 // tile (mt, nt) = (t / 4, t % 4)
                aH[l - 1][i] = dw_tile(Ds, C::LDH, t / 4, Hs, C::LDH, t % 4, aH[l - 1][i]);
            }
            h4 dn[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const f4 acc = mfma_ksteps<4>([&](int kt) { return lds_tr_read(Wl + C::WH_OFF + ((l - 1) * 64 + kt * 16 + 4 * g + tq) * C::LDH + mt * 16 + 4 * tp); },   // W_l^T
                                              d, f4{0, 0, 0, 0});
#pragma unroll
                for (int r = 0; r < 4; r++) dn[mt][r] = ((float)h[l - 1][mt][r] > 0.0f) ? (half_t)acc[r] : (half_t)0.0f;
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) d[mt] = dn[mt];
        }
        // ---- input layer: dW0 tiles dealt to the waves ; own rows: dX = W0^T dH_0
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; mt++) *reinterpret_cast<h4*>(Ds + (r0 + c) * C::LDH + mt * 16 + 4 * g) = d[mt];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < C::T0; i++) {
            const int t = w + i * WAVES;                     // tile (mt, nt) = (t / KT0, t % KT0)
            if (t < 4 * KT0) a0[i] = dw_tile(Ds, C::LDH, t / KT0, Xs, C::LDX, t % KT0, a0[i]);
        }
        if constexpr (MODE == 1) {
            const f4 acc = mfma_ksteps<4>([&](int kt) { return lds_tr_read(Wl + C::W0_OFF + (kt * 16 + 4 * g + tq) * C::LDX + 16 + 4 * tp); },   // W0^T, features 16..31
                                          d, f4{0, 0, 0, 0});
            const half_t v3 = (half_t)acc[3];
            const half_t prev = __builtin_bit_cast(half_t, (uint16_t)__shfl_up((int)__builtin_bit_cast(uint16_t, v3), 16, 64));
            h4 v;
            v[0] = prev;
            if (g == 0) v[0] = (half_t)(ha.grad_sigmas[row] * ha.density_scale * expf(lae::clampf((float)hq[0], -15.0f, 15.0f)));
            v[1] = (half_t)acc[0]; v[2] = (half_t)acc[1]; v[3] = (half_t)acc[2];
            if (active) *reinterpret_cast<h4*>(grad_in + row * 16 + 4 * g) = v;
        } else if (grad_in) {
#pragma unroll
            for (int it = 0; it < KT0; it++) {
                const f4 acc = mfma_ksteps<4>([&](int kt) { return lds_tr_read(Wl + C::W0_OFF + (kt * 16 + 4 * g + tq) * C::LDX + it * 16 + 4 * tp); },   // W0^T
                                              d, f4{0, 0, 0, 0});
                h4 v;
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = (half_t)acc[r];
                if (!active) continue;
                if (KT0 == 2 && ha.level_major) {
                    const uint2 u = __builtin_bit_cast(uint2, v);
                    const size_t Bn = (size_t)n_tiles * 16;
                    *reinterpret_cast<uint32_t*>(grad_in + ((size_t)(8 * it + 2 * g) * Bn + row) * 2) = u.x;
                    *reinterpret_cast<uint32_t*>(grad_in + ((size_t)(8 * it + 2 * g + 1) * Bn + row) * 2) = u.y;
                } else
                    *reinterpret_cast<h4*>(grad_in + row * IN + it * 16 + 4 * g) = v;
            }
        }
    }

    // ---- every wave owns its tiles: straight to this workgroup's slab (C/D layout: lane (c, g), reg r = element [4g + r][c])
    float* slab = slabs + (size_t)blockIdx.x * nW;
    auto store_tile = [&](size_t base, int ld, const f4& v) {        // element [rr][cc] of the tile -> base + rr * ld + cc
#pragma unroll
        for (int r = 0; r < 4; r++) slab[base + (size_t)(4 * g + r) * ld + c] = v[r];
    };
#pragma unroll
    for (int i = 0; i < C::T0; i++) {
        const int t = w + i * WAVES;
        if (t < 4 * KT0) store_tile((size_t)((t / KT0) * 16) * IN + (t % KT0) * 16, IN, a0[i]);
    }
#pragma unroll
    for (int m = 0; m < NH; m++)
#pragma unroll
        for (int i = 0; i < C::TH; i++) {
            const int t = w + i * WAVES;
            store_tile((size_t)64 * IN + (size_t)m * 4096 + (size_t)((t / 4) * 16) * 64 + (t % 4) * 16, 64, aH[m][i]);
        }
    if (w < 4) store_tile((size_t)64 * IN + (size_t)NH * 4096 + w * 16, 64, aO);
}

// in-kernel phase stamps for tools/ubench/mlp_probe.hip (compiled in only there): 100 MHz wall clock + shader clock
#ifdef LAE_MLP_STAMPS
__device__ unsigned long long g_mlp_stamps[4096 * 64];
#define MLP_STAMP(slot, i) do { if ((threadIdx.x & 63) == 0 && (i) < 32) { g_mlp_stamps[(size_t)(slot) * 64 + 2 * (i)] = wall_clock64(); \
                                g_mlp_stamps[(size_t)(slot) * 64 + 2 * (i) + 1] = __builtin_readcyclecounter(); } } while (0)
#define MLP_PHASE(slot, i) do { if (stamp_i == 3) MLP_STAMP(slot, 16 + (i)); } while (0)
#else
#define MLP_STAMP(slot, i) do { } while (0)
#define MLP_PHASE(slot, i) do { } while (0)
#endif
// ---------------------------------------------------------------- fused backward, round-3 form: no activation ever touches LDS
// k_mlp_bwd_coop shares X / H / D / G tiles of 128 rows among 8 waves: 7 workgroup barriers per 128 rows and every
// activation written row-major to LDS and read back transposed (66 % of its wave cycles parked at waitcnt / barrier, 10 % of
// the MFMA rate).  What the weight gradient needs is the TRANSPOSE of tiles that the chain already holds in registers:
//   chain layout (C/D of W x act^T):  lane (c, g) holds feats 4g .. 4g+3 of batch row c
//   dW operand layout              :  lane (c, g) holds batch rows 4g .. 4g+3 of feature c     (A = D^T and B = H of dW = D^T H)
// and a 16 x 16 transpose IS one MFMA: with the chain-layout registers as the A operand and the identity as B,
// C[m][n] = sum_k A[m][k] I[k][n] = A[m][n] comes out in the C/D layout = lane (n = feature, rows 4g..) -- exact (products
// with 0 / 1, fp32 accumulate, values are halves).  So a wave keeps its own two 16-row tiles in registers through recompute,
// dL/dH chain and transposes, accumulates ALL dW tiles of the net itself with K = 32 MFMAs over the 32 rows (tile A rows in
// the low k-slots, tile B rows in the high ones, the same permutation in both operands), and never synchronises with another
// wave until the final fixed-order reduction of the four waves' dW through LDS.  Only the weights are read from LDS.
// MFMAs per 16 rows (colour net): recompute 20, chain 22, transposes 27, dW 22 = 91 (coop: 66) -- the matrix pipe was never
// the limit.
template <int IN, int NH>
struct WaveCfg {
    static constexpr int KT0 = IN / 16;
    static constexpr int LDX = IN + 8, LDH = 72;
    static constexpr int W0_OFF = 0, WH_OFF = 64 * LDX, WO_OFF = WH_OFF + NH * 64 * LDH, W_HALVES = WO_OFF + 16 * LDH;
    static constexpr int N_TILES = 4 * KT0 + NH * 16 + 4;                 // 16x16 dW tiles: W0 | hidden | Wout
    static constexpr size_t LDS_BYTES = (size_t)W_HALVES * 2 + (size_t)N_TILES * 2048;      // weight image + two dW images (final reduction)
};

// 16 x 16 transpose on the matrix core: chain layout -> dW operand layout (see above)
__device__ __forceinline__ h4 mfma_transpose(const h4& x, const h4& ident) {
    const f4 r = mfma16(x, ident, f4{0, 0, 0, 0});
    h4 o;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (half_t)r[i];
    return o;
}
// dL/dH of a ReLU layer: acc where the (post-ReLU, hence >= +0) activation is positive, else 0; packed halves.  -h as int16
// is negative exactly when h > 0; its sign bit smeared over the half is the mask.  (Written with vector types the compiler
// turns this back into four compares and selects per pair.)
__device__ __forceinline__ h4 relu_bwd4(const f4& acc, const h4& h) {
    h2 lo = {(half_t)acc[0], (half_t)acc[1]}, hi = {(half_t)acc[2], (half_t)acc[3]};
    const uint2 hb = __builtin_bit_cast(uint2, h);
    uint32_t d0 = __builtin_bit_cast(uint32_t, lo), d1 = __builtin_bit_cast(uint32_t, hi), m0, m1;
    const uint32_t fifteen = 0x000f000fu;
    asm("v_pk_sub_i16 %0, 0, %1\n\tv_pk_ashrrev_i16 %0, %2, %0" : "=&v"(m0) : "v"(hb.x), "v"(fifteen));
    asm("v_pk_sub_i16 %0, 0, %1\n\tv_pk_ashrrev_i16 %0, %2, %0" : "=&v"(m1) : "v"(hb.y), "v"(fifteen));
    d0 &= m0; d1 &= m1;
    return __builtin_bit_cast(h4, uint2{d0, d1});
}

// four 16 x 16 transposes at once: the MFMAs first, the conversions after (a conversion right behind its MFMA waits for it)
template <int N>
__device__ __forceinline__ void mfma_transpose_n(const h4 (&x)[N], const h4& ident, h4 (&o)[N]) {
    f4 r[N];
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = mfma16(x[i], ident, f4{0, 0, 0, 0});
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) o[i][j] = (half_t)r[i][j];
}

// WPE = waves per SIMD the kernel is compiled for: 1 (up to 512 registers: the colour net's 176 accumulator registers + two tiles
// of working state) or 2 (256 registers: nets with one hidden GEMM; the inputs are then not requested ahead -- the partner wave
// covers the latency -- which frees the registers of the second request)
template <int IN, int NH, int MODE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_mlp_bwd_wave(
    const half_t* __restrict__ grad, const half_t* __restrict__ x, const half_t* __restrict__ W, uint32_t n_tiles,
    half_t* __restrict__ grad_in, float* __restrict__ slabs, uint32_t nW, HeadBwdArgs ha) {
    using C = WaveCfg<IN, NH>;
    constexpr int KT0 = C::KT0;
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* Wl = lds;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tq = (lane & 15) >> 2, tp = lane & 3;           // transposed-read lane address (ds_read_tr16_b64)
    const h4 zero4 = h4{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
    MLP_STAMP(blockIdx.x * 4 + w, 0);

    const uint32_t n_pairs = (n_tiles + 1) / 2;
    const uint32_t wave0 = blockIdx.x * 4 + (uint32_t)w, nwaves = gridDim.x * 4;
    const size_t Bn = (size_t)n_tiles * 16;
    // ---- inputs, requested one pair ahead (one wave per SIMD: nothing else hides a memory latency per pair)
    struct Req { h4 xf[2][KT0]; h4 gf[2]; h4 hq[2]; float dir[2][3], y[2][3], gy[2][3], gs[2]; };
    constexpr bool AHEAD = WPE == 1;
    Req nx = {};                                                         // (copied whole below: no member may be indeterminate)
    auto request = [&](uint32_t pair) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t tile = min(2 * pair + t, n_tiles - 1);          // a tile past the batch re-reads the last one; its operands are zeroed below
            const size_t row = (size_t)tile * 16 + c;
            if constexpr (MODE == 1) {
                nx.hq[t] = *reinterpret_cast<const h4*>(x + row * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 3; r++) nx.dir[t][r] = ha.dirs[3 * row + r];
                if (g == 0) {
#pragma unroll
                    for (int r = 0; r < 3; r++) { nx.y[t][r] = ha.rgbs[row * 3 + r]; nx.gy[t][r] = ha.grad_rgbs[row * 3 + r]; }
                    nx.gs[t] = ha.grad_sigmas[row];
                }
            } else {
                if constexpr (KT0 == 2) load_enc_frags(x, row, Bn, g, ha.level_major, nx.xf[t]);
                else {
#pragma unroll
                    for (int kt = 0; kt < KT0; kt++) nx.xf[t][kt] = *reinterpret_cast<const h4*>(x + row * IN + kt * 16 + 4 * g);
                }
                nx.gf[t] = *reinterpret_cast<const h4*>(grad + row * 16 + 4 * g);
            }
        }
    };
    if (AHEAD && wave0 < n_pairs) request(wave0);

    // ---- weights -> LDS (row-major, padded): every 16-byte global load of the workgroup in flight before the first store
    {
        constexpr int CH0 = 64 * IN / 8, CHH = NH * 64 * 64 / 8, CHO = 16 * 64 / 8;
        constexpr int N0 = (CH0 + 255) / 256, N1 = (CHH + 255) / 256, N2 = (CHO + 255) / 256;
        uint4 b0[N0], b1[N1], b2[N2];
#pragma unroll
        for (int i = 0; i < N0; i++) { const int e = (int)threadIdx.x + 256 * i; if (e < CH0) b0[i] = *reinterpret_cast<const uint4*>(W + (size_t)e * 8); }
#pragma unroll
        for (int i = 0; i < N1; i++) { const int e = (int)threadIdx.x + 256 * i; if (e < CHH) b1[i] = *reinterpret_cast<const uint4*>(W + 64 * IN + (size_t)e * 8); }
#pragma unroll
        for (int i = 0; i < N2; i++) { const int e = (int)threadIdx.x + 256 * i; if (e < CHO) b2[i] = *reinterpret_cast<const uint4*>(W + 64 * IN + NH * 4096 + (size_t)e * 8); }
#pragma unroll
        for (int i = 0; i < N0; i++) { const int e = (int)threadIdx.x + 256 * i; if (e < CH0) { const int r = (e * 8) / IN, k = (e * 8) % IN; *reinterpret_cast<uint4*>(Wl + C::W0_OFF + r * C::LDX + k) = b0[i]; } }
#pragma unroll
        for (int i = 0; i < N1; i++) { const int e = (int)threadIdx.x + 256 * i; if (e < CHH) { const int r = (e * 8) / 64, k = (e * 8) % 64; *reinterpret_cast<uint4*>(Wl + C::WH_OFF + r * C::LDH + k) = b1[i]; } }
#pragma unroll
        for (int i = 0; i < N2; i++) { const int e = (int)threadIdx.x + 256 * i; if (e < CHO) { const int r = (e * 8) / 64, k = (e * 8) % 64; *reinterpret_cast<uint4*>(Wl + C::WO_OFF + r * C::LDH + k) = b2[i]; } }
    }
    __syncthreads();

    h4 ident;
#pragma unroll
    for (int j = 0; j < 4; j++) ident[j] = (4 * g + j == c) ? (half_t)1.0f : (half_t)0.0f;

    f4 dW0[4][KT0], dWh[NH][4][4], dWo[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < KT0; j++) dW0[i][j] = f4{0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < NH; m++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) dWh[m][i][j] = f4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) dWo[j] = f4{0, 0, 0, 0};

    [[maybe_unused]] int stamp_i = 1;
    for (uint32_t pair = wave0; pair < n_pairs; pair += nwaves) {
        MLP_STAMP(blockIdx.x * 4 + w, stamp_i); stamp_i++;
        // ---- this pair's inputs (chain layout) from the request issued one iteration ago; then the next request
        if (!AHEAD) request(pair);
        const Req cur = nx;
        if (AHEAD && pair + nwaves < n_pairs) request(pair + nwaves);
        h4 xf[2][KT0], gf[2], hq[2];
        bool act[2];
        size_t row[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t tile = 2 * pair + t;
            act[t] = tile < n_tiles;                                     // wave-uniform
            row[t] = (size_t)(act[t] ? tile : 0) * 16 + c;
            gf[t] = zero4; hq[t] = cur.hq[t];
            if constexpr (MODE == 1) {
                static_assert(MODE == 0 || KT0 == 2, "head colour net has a 32-wide input");
                color_inputs(cur.dir[t][0], cur.dir[t][1], cur.dir[t][2], hq[t], g, xf[t]);
                if (g == 0) {
#pragma unroll
                    for (int r = 0; r < 3; r++) gf[t][r] = (half_t)(cur.gy[t][r] * cur.y[t][r] * (1.0f - cur.y[t][r]));
                }
            } else {
#pragma unroll
                for (int kt = 0; kt < KT0; kt++) xf[t][kt] = cur.xf[t][kt];
                gf[t] = cur.gf[t];
            }
            // a tile past the batch (odd tile count) re-reads the last tile and contributes nothing: with dL/dout = 0 every dL/dH
            // and so every dW term of the tile is zero; its dX is not stored
            if (!act[t]) gf[t] = zero4;
        }
        MLP_PHASE(blockIdx.x * 4 + w, 0);
        // ---- recompute the hidden activations (post-ReLU) of both tiles; one fragment read serves both.  The fragments of a
        // layer are requested one layer AHEAD (with one wave per SIMD nothing else covers an LDS latency: 42 % of the wave's
        // cycles were s_waitcnt, profiles/r3a_sq_stall_breakdown.txt) and pinned there with a scheduling barrier.
        h4 h[2][NH + 1][4];
        {
            auto load_w0 = [&](h4 (&a)[4][KT0]) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int kt = 0; kt < KT0; kt++) a[mt][kt] = *reinterpret_cast<const h4*>(Wl + C::W0_OFF + (mt * 16 + c) * C::LDX + kt * 16 + 4 * g);
            };
            auto load_wh = [&](int l, h4 (&a)[4][4]) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int kt = 0; kt < 4; kt++) a[mt][kt] = *reinterpret_cast<const h4*>(Wl + C::WH_OFF + (l * 64 + mt * 16 + c) * C::LDH + kt * 16 + 4 * g);
            };
            f4 acc[2][4];
            h4 a0[4][KT0], a1[4][4], a2[4][4];
            load_w0(a0);
            load_wh(0, a1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int t = 0; t < 2; t++) acc[t][mt] = mfma_ksteps<KT0>([&](int kt) { return a0[mt][kt]; }, xf[t], f4{0, 0, 0, 0});
            if constexpr (NH > 1) load_wh(1, a2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; t++) relu4(acc[t], h[t][0]);
#pragma unroll
            for (int l = 0; l < NH; l++) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int t = 0; t < 2; t++)
                        acc[t][mt] = mfma_ksteps<4>([&](int kt) { return (l & 1) ? a2[mt][kt] : a1[mt][kt]; }, h[t][l], f4{0, 0, 0, 0});
                static_assert(NH <= 2, "fragment double buffer: two hidden GEMMs at most");
#pragma unroll
                for (int t = 0; t < 2; t++) relu4(acc[t], h[t][l + 1]);
            }
        }
        MLP_PHASE(blockIdx.x * 4 + w, 1);
        // transposed weight fragments (A[f][o] = W[o][f]) of the chain, likewise requested one step ahead
        auto load_wt = [&](int l, h4 (&a)[4][4]) {               // W_l^T of hidden GEMM l (1-based)
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int kt = 0; kt < 4; kt++) a[mt][kt] = lds_tr_read(Wl + C::WH_OFF + ((l - 1) * 64 + kt * 16 + 4 * g + tq) * C::LDH + mt * 16 + 4 * tp);
        };
        h4 wt[2][4][4];                                         // double buffer: hidden layer l uses wt[l & 1]
        h4 wo[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) wo[mt] = lds_tr_read(Wl + C::WO_OFF + (4 * g + tq) * C::LDH + mt * 16 + 4 * tp);            // A[f][o] = Wout[o][f]
        load_wt(NH, wt[NH & 1]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- output layer: dWout += G^T H_NH over the 32 rows ; dH_NH = (Wout^T G) * relu'
        h4 d[2][4];
        {
            h4 TG[2], TH[2][4];
            { const h4 gg[2] = {gf[0], gf[1]}; mfma_transpose_n<2>(gg, ident, TG); }
#pragma unroll
            for (int t = 0; t < 2; t++) mfma_transpose_n<4>(h[t][NH], ident, TH[t]);
            f4 da[2][4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int t = 0; t < 2; t++) da[t][mt] = mfma16(wo[mt], gf[t], f4{0, 0, 0, 0});
#pragma unroll
            for (int nt = 0; nt < 4; nt++) dWo[nt] = mfma32(TG[0], TG[1], TH[0][nt], TH[1][nt], dWo[nt]);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++) d[t][mt] = relu_bwd4(da[t][mt], h[t][NH][mt]);
        }
        MLP_PHASE(blockIdx.x * 4 + w, 2);
        // ---- hidden layers, last to first: dW_l += D_l^T H_{l-1} ; dH_{l-1} = (W_l^T dH_l) * relu'
        h4 wx[MODE == 1 ? 1 : KT0][4];                          // W0^T fragments of the dX product (MODE 1: features 16..31 only)
#pragma unroll
        for (int l = NH; l >= 1; l--) {
            if (l > 1) load_wt(l - 1, wt[(l - 1) & 1]);          // next step's fragments
            else {
#pragma unroll
                for (int it = 0; it < (MODE == 1 ? 1 : KT0); it++)
#pragma unroll
                    for (int kt = 0; kt < 4; kt++)
                        wx[it][kt] = lds_tr_read(Wl + C::W0_OFF + (kt * 16 + 4 * g + tq) * C::LDX + (MODE == 1 ? 16 : it * 16) + 4 * tp);   // W0^T
            }
            __builtin_amdgcn_sched_barrier(0);
            h4 TD[2][4], TH[2][4];
#pragma unroll
            for (int t = 0; t < 2; t++) { mfma_transpose_n<4>(d[t], ident, TD[t]); mfma_transpose_n<4>(h[t][l - 1], ident, TH[t]); }
            f4 da[2][4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int t = 0; t < 2; t++) da[t][mt] = mfma_ksteps<4>([&](int kt) { return wt[l & 1][mt][kt]; }, d[t], f4{0, 0, 0, 0});
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) dWh[l - 1][mt][nt] = mfma32(TD[0][mt], TD[1][mt], TH[0][nt], TH[1][nt], dWh[l - 1][mt][nt]);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++) d[t][mt] = relu_bwd4(da[t][mt], h[t][l - 1][mt]);
        }
        MLP_PHASE(blockIdx.x * 4 + w, 3);
        // ---- input layer: dW0 += D_0^T X ; dX = W0^T dH_0
        {
            h4 TD[2][4], TX[2][KT0];
#pragma unroll
            for (int t = 0; t < 2; t++) { mfma_transpose_n<4>(d[t], ident, TD[t]); mfma_transpose_n<KT0>(xf[t], ident, TX[t]); }
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int nt = 0; nt < KT0; nt++) dW0[mt][nt] = mfma32(TD[0][mt], TD[1][mt], TX[0][nt], TX[1][nt], dW0[mt][nt]);
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f4 acc = mfma_ksteps<4>([&](int kt) { return wx[0][kt]; }, d[t], f4{0, 0, 0, 0});
                const half_t v3 = (half_t)acc[3];
                const half_t prev = __builtin_bit_cast(half_t, (uint16_t)__shfl_up((int)__builtin_bit_cast(uint16_t, v3), 16, 64));
                h4 v;
                v[0] = prev;
                if (g == 0) v[0] = (half_t)(cur.gs[t] * ha.density_scale * expf(lae::clampf((float)hq[t][0], -15.0f, 15.0f)));
                v[1] = (half_t)acc[0]; v[2] = (half_t)acc[1]; v[3] = (half_t)acc[2];
                if (act[t]) *reinterpret_cast<h4*>(grad_in + row[t] * 16 + 4 * g) = v;
            }
        } else if (grad_in) {
#pragma unroll
            for (int it = 0; it < KT0; it++) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const f4 acc = mfma_ksteps<4>([&](int kt) { return wx[MODE == 1 ? 0 : it][kt]; }, d[t], f4{0, 0, 0, 0});
                    h4 v;
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = (half_t)acc[r];
                    if (!act[t]) continue;
                    if (KT0 == 2 && ha.level_major) {
                        const uint2 u = __builtin_bit_cast(uint2, v);
                        *reinterpret_cast<uint32_t*>(grad_in + ((size_t)(8 * it + 2 * g) * Bn + row[t]) * 2) = u.x;
                        *reinterpret_cast<uint32_t*>(grad_in + ((size_t)(8 * it + 2 * g + 1) * Bn + row[t]) * 2) = u.y;
                    } else
                        *reinterpret_cast<h4*>(grad_in + row[t] * IN + it * 16 + 4 * g) = v;
                }
            }
        }
        MLP_PHASE(blockIdx.x * 4 + w, 4);
    }

    // ---- the 4 waves' dW tiles -> one fp32 slab per workgroup, in a fixed order: (w0 + w2) + (w1 + w3).  Tiles travel in
    // the C/D register layout (one 16-byte LDS access per tile and lane); the slab store undoes it.
    MLP_STAMP(blockIdx.x * 4 + w, 14);
    f4* red = reinterpret_cast<f4*>(lds + C::W_HALVES);                  // [2][N_TILES][64] f4
    auto for_tiles = [&](auto&& fn) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < KT0; j++) fn(t++, dW0[i][j]);
#pragma unroll
        for (int m = 0; m < NH; m++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) fn(t++, dWh[m][i][j]);
#pragma unroll
        for (int j = 0; j < 4; j++) fn(t++, dWo[j]);
    };
    if (w < 2) for_tiles([&](int t, const f4& v) { red[((size_t)w * C::N_TILES + t) * 64 + lane] = v; });
    __syncthreads();
    if (w >= 2) for_tiles([&](int t, const f4& v) { f4& r = red[((size_t)(w - 2) * C::N_TILES + t) * 64 + lane]; r = r + v; });
    __syncthreads();
    float* slab = slabs + (size_t)blockIdx.x * nW;
    for (uint32_t e = threadIdx.x; e < (uint32_t)C::N_TILES * 64; e += 256) {
        const uint32_t t = e >> 6, ln = e & 63, cc = ln & 15, gg = ln >> 4;
        const f4 v = red[e] + red[(size_t)C::N_TILES * 64 + e];
        size_t base; uint32_t ld;
        if (t < 4u * KT0) { base = (size_t)((t / KT0) * 16) * IN + (t % KT0) * 16; ld = IN; }
        else if (t < 4u * KT0 + NH * 16u) {
            const uint32_t u = t - 4 * KT0, m = u / 16, i = (u % 16) / 4, j = u % 4;
            base = (size_t)64 * IN + (size_t)m * 4096 + (size_t)(i * 16) * 64 + j * 16; ld = 64;
        } else { base = (size_t)64 * IN + (size_t)NH * 4096 + (t - 4 * KT0 - NH * 16) * 16; ld = 64; }
#pragma unroll
        for (int r = 0; r < 4; r++) slab[base + (size_t)(4 * gg + r) * ld + cc] = v[r];
    }
    MLP_STAMP(blockIdx.x * 4 + w, 15);
}

// 2 = wave-private dW with MFMA transposes (k_mlp_bwd_wave, default), 0 = workgroup-cooperative dW (k_mlp_bwd_coop, round 2), 3 = wave-private
// for one hidden GEMM and cooperative for two; A/B switch, see lae_ffmlp_set_mode
int g_bwd_fused_variant = -1;
static int bwd_fused_variant() {
    if (g_bwd_fused_variant < 0) { const char* e = getenv("LAE_MLP_BWD_VARIANT"); g_bwd_fused_variant = e ? atoi(e) : 2; if (g_bwd_fused_variant != 0 && g_bwd_fused_variant != 3) g_bwd_fused_variant = 2; }
    return g_bwd_fused_variant;
}

template <int IN, int NH, int MODE = 0>
int launch_bwd_fused(const half_t* grad, const half_t* x, const half_t* W, uint32_t B, half_t* grad_in, half_t* gw, hipStream_t s,
                     HeadBwdArgs ha = HeadBwdArgs{}, int accumulate = 0, float* slabs = nullptr, uint32_t* n_slices_out = nullptr,
                     int32_t* nf_flag = nullptr) {
    // slabs != nullptr: the partial slabs go to the caller's region (room for 2 * num_cus slices) and the reduction is
    // left to the caller (lae_nerf_head_backward reduces both networks in one launch)
    const uint32_t nW = 64 * (IN + 64 * NH + 16);
    const uint32_t n_tiles = B / 16;
    uint32_t blocks;
    float* ws;
    const int variant = bwd_fused_variant() == 3 ? (NH == 1 ? 2 : 0) : bwd_fused_variant();      // 3: wave-private for one hidden GEMM, cooperative for two
    if (variant == 2) {
        using C = WaveCfg<IN, NH>;
        constexpr int WPE = 1;       // 2 waves per SIMD measured on the sigma net (26 spilled registers): 3.1 us per pair and SIMD against 2.4
        const size_t lds_bytes = C::LDS_BYTES;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_bwd_wave<IN, NH, MODE, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes) != hipSuccess) return LAE_ELAUNCH;
            attr_set = true;
        }
        blocks = std::max(1u, std::min(lae::cdiv(lae::cdiv(n_tiles, 2), 4), (uint32_t)lae::num_cus() * WPE));
        ws = slabs ? slabs : reinterpret_cast<float*>(lae::workspace(lae::WS_FFMLP_SLABS, (size_t)blocks * nW * sizeof(float), s));
        if (!ws) return LAE_ELAUNCH;
        k_mlp_bwd_wave<IN, NH, MODE, WPE><<<blocks, 256, lds_bytes, s>>>(grad, x, W, n_tiles, grad_in, ws, nW, ha);
    } else {                                               // the round-2 kernel, kept as the A/B predecessor
        constexpr int WAVES = 8;
        using C = CoopCfg<IN, NH, WAVES>;
        const size_t lds_bytes = (size_t)C::LDS_HALVES * 2;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_bwd_coop<IN, NH, MODE, WAVES>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return LAE_ELAUNCH;
            attr_set = true;
        }
        blocks = std::max(1u, std::min(lae::cdiv(n_tiles, WAVES), (uint32_t)lae::num_cus() * 2));
        ws = slabs ? slabs : reinterpret_cast<float*>(lae::workspace(lae::WS_FFMLP_SLABS, (size_t)blocks * nW * sizeof(float), s));
        if (!ws) return LAE_ELAUNCH;
        k_mlp_bwd_coop<IN, NH, MODE, WAVES><<<blocks, 64 * WAVES, lds_bytes, s>>>(grad, x, W, n_tiles, grad_in, ws, nW, ha);
    }
    if (n_slices_out) *n_slices_out = blocks;
    if (!slabs) k_dw_reduce<<<lae::cdiv(nW, 64), 64 * DWR_GROUPS, 0, s>>>(ws, blocks, nW, gw, accumulate, nf_flag);
    return LAE_OK;
}

struct Head4Scratch { half_t sh[64][16]; float q[64][4]; };             // per wave: SH block and logits on their way between row-per-lane and fragment layout

// ---------------------------------------------------------------- fused NeRF head, the kernel that ships
// (round 1-2: one 16-row tile per wave iteration, weights row-major in LDS; round 3 first form: fragments in 144 registers,
// one wave per SIMD -- both removed in round 4, their measurements are in DESIGN.md appendix A.)
//  * a wave owns FOUR tiles (64 rows) per iteration and every layer is issued for all four before its activations are
//    needed: four independent MFMA -> cvt / relu -> MFMA chains per wave instead of one;
//  * per-row scalar work (SH polynomials, exp, three sigmoids with an IEEE division each) is done by lane L for row L once,
//    not by the four lane groups of a 16-row tile; the values reach / leave the MFMA fragment layout through a wave-private
//    3 KB LDS scratch (2 wide stores + 4 fragment reads for the SH block; one masked store per tile + one read for the
//    density logit and the colour logits).  sigma / rgb leave as one coalesced row-per-lane store;
//  * the A fragments are re-read from a pre-swizzled LDS image (one conflict-free ds_read_b128 per fragment PAIR and four
//    tiles) instead of being held in registers: in-kernel stamps (tools/ubench/mlp_probe.hip) showed that 1024 waves each
//    pulling 37 KB of fragments from L2 take 7.5 us before the first row is touched, and that one wave alone on a SIMD
//    issues an instruction every ~5 cycles whatever it is -- the rate of this kernel is instructions per row x issue
//    interval, and the issue interval halves with a second wave on the SIMD.
// LDS image of a [16 MT, 32 KP] weight matrix: fragment pair (mt, p) of lane l = 16 bytes at ((mt * KP + p) * 64 + l) * 16:
// halves 0-3 = W[16 mt + c][32 p + 4 g ..], halves 4-7 = W[16 mt + c][32 p + 16 + 4 g ..] (the two operands of one K = 32 MFMA).
template <int MT, int KP, int NTH>
__device__ __forceinline__ void stage_swizzled_issue(const half_t* __restrict__ W, uint4 (&buf)[(MT * KP * 64 + NTH - 1) / NTH]) {
    constexpr int CHUNKS = MT * 16 * KP * 32 / 8;                        // 16-byte chunks of the row-major matrix
#pragma unroll
    for (int i = 0; i < (MT * KP * 64 + NTH - 1) / NTH; i++) {
        const int e = (int)threadIdx.x + i * NTH;
        if (e < CHUNKS) buf[i] = *reinterpret_cast<const uint4*>(W + (size_t)e * 8);
    }
}
template <int MT, int KP, int NTH>
__device__ __forceinline__ void stage_swizzled_store(half_t* img, const uint4 (&buf)[(MT * KP * 64 + NTH - 1) / NTH]) {
    constexpr int CHUNKS = MT * 16 * KP * 32 / 8, K = KP * 32;
#pragma unroll
    for (int i = 0; i < (MT * KP * 64 + NTH - 1) / NTH; i++) {
        const int e = (int)threadIdx.x + i * NTH;
        if (e < CHUNKS) {
            const int row = (e * 8) / K, k = (e * 8) % K;                // halves k .. k + 7 of row `row`
            const int mt = row >> 4, cc = row & 15;
#pragma unroll
            for (int piece = 0; piece < 2; piece++) {
                const int kk = k + 4 * piece, p = kk >> 5, hh = (kk >> 4) & 1, gg = (kk >> 2) & 3;
                const uint2 v = piece ? uint2{buf[i].z, buf[i].w} : uint2{buf[i].x, buf[i].y};
                *reinterpret_cast<uint2*>(img + ((size_t)((mt * KP + p) * 64 + gg * 16 + cc)) * 8 + hh * 4) = v;
            }
        }
    }
}
struct Head5Img {                                                       // halves
    static constexpr int S0 = 0, S1 = S0 + 64 * 32, SO = S1 + 64 * 64, C0 = SO + 16 * 64, C1 = C0 + 64 * 32, C2 = C1 + 64 * 64,
                         CO = C2 + 64 * 64, END = CO + 16 * 64;
};
// 64-wide layer on NT tiles, A fragment pairs from the swizzled image
template <int KP, int NT>
__device__ __forceinline__ void layer64s(const half_t* img, int lane, const h4 (&in)[NT][2 * KP], h4 (&out)[NT][4]) {
    f4 acc[NT][4];
#pragma unroll
    for (int p = 0; p < KP; p++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const h8 a = *reinterpret_cast<const h8*>(img + ((size_t)((mt * KP + p) * 64 + lane)) * 8);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const h8 b = __builtin_shufflevector(in[t][2 * p], in[t][2 * p + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, p == 0 ? f4{0, 0, 0, 0} : acc[t][mt], 0, 0, 0);
            }
        }
#pragma unroll
    for (int t = 0; t < NT; t++) relu4(acc[t], out[t]);
}
template <int NT>
__device__ __forceinline__ void out16s(const half_t* img, int lane, const h4 (&in)[NT][4], f4 (&o)[NT]) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const h8 a = *reinterpret_cast<const h8*>(img + ((size_t)(p * 64 + lane)) * 8);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const h8 b = __builtin_shufflevector(in[t][2 * p], in[t][2 * p + 1], 0, 1, 2, 3, 4, 5, 6, 7);
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, p == 0 ? f4{0, 0, 0, 0} : o[t], 0, 0, 0);
        }
    }
}

template <bool COLOR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nerf_head_fwd5(
    const half_t* __restrict__ enc, const float* __restrict__ dirs, const half_t* __restrict__ Ws, const half_t* __restrict__ Wc,
    uint32_t n_tiles, float density_scale, half_t* __restrict__ h_out, float* __restrict__ sigmas, float* __restrict__ rgbs,
    int level_major, const uint32_t* __restrict__ n_rows_dev, uint32_t lm_rows) {
    constexpr int NT = 4, NTH = 64 * WAVES;
    using I = Head5Img;
    if (n_rows_dev) n_tiles = min(n_tiles, (*n_rows_dev + 15u) / 16u);
    if (n_tiles == 0) return;
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* img = lds;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Head4Scratch* sc = reinterpret_cast<Head4Scratch*>(lds + (COLOR ? I::END : I::C0)) + w;
    MLP_STAMP(blockIdx.x * WAVES + w, 0);
    // ---- weights -> swizzled LDS image: every global load of the workgroup in flight before the first LDS store
    {
        constexpr int N1 = (4 * 1 * 64 + NTH - 1) / NTH, N2 = (4 * 2 * 64 + NTH - 1) / NTH, NO = (1 * 2 * 64 + NTH - 1) / NTH;
        uint4 b0[N1], b1[N2], b2[NO];
        stage_swizzled_issue<4, 1, NTH>(Ws, b0);
        stage_swizzled_issue<4, 2, NTH>(Ws + 64 * 32, b1);
        stage_swizzled_issue<1, 2, NTH>(Ws + 64 * 32 + 4096, b2);
        if constexpr (COLOR) {
            uint4 c0[N1], c1[N2], c2[N2], c3[NO];
            stage_swizzled_issue<4, 1, NTH>(Wc, c0);
            stage_swizzled_issue<4, 2, NTH>(Wc + 64 * 32, c1);
            stage_swizzled_issue<4, 2, NTH>(Wc + 64 * 32 + 4096, c2);
            stage_swizzled_issue<1, 2, NTH>(Wc + 64 * 32 + 8192, c3);
            stage_swizzled_store<4, 1, NTH>(img + I::C0, c0);
            stage_swizzled_store<4, 2, NTH>(img + I::C1, c1);
            stage_swizzled_store<4, 2, NTH>(img + I::C2, c2);
            stage_swizzled_store<1, 2, NTH>(img + I::CO, c3);
        }
        stage_swizzled_store<4, 1, NTH>(img + I::S0, b0);
        stage_swizzled_store<4, 2, NTH>(img + I::S1, b1);
        stage_swizzled_store<1, 2, NTH>(img + I::SO, b2);
    }
    const uint32_t n_groups = (n_tiles + NT - 1) / NT;
    const uint32_t wave0 = blockIdx.x * WAVES + (uint32_t)w, nwaves = gridDim.x * WAVES;
    const uint32_t n_rows = n_tiles * 16u;
    h4 xf_n[NT][2] = {};
    float d_n[3] = {0.f, 0.f, 0.f};
    // inputs through buffer descriptors: one 32-bit lane offset per group, the tile / k-step / level strides ride in the
    // instruction's immediate and scalar offsets (the 64-bit address arithmetic of 16 + 3 plain loads was ~90 instructions
    // per group), and the hardware range check returns zeros past the end instead of branches around the loads.  Level-major
    // features: a row past the live rows of plane l reads plane l + 1 (memory of the same buffer; its results are never stored).
    const __amdgpu_buffer_rsrc_t rs_enc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(enc), 0,
                                                                            (int)((level_major ? lm_rows : n_rows) * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dir = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(COLOR ? dirs : sigmas), 0, (int)(n_rows * 12u), 0x00020000);
    const uint32_t plane = lm_rows * 4u;                               // bytes per level plane
    auto request = [&](uint32_t grp) {
        const uint32_t row0 = grp * 64u;
        if (level_major) {
            const uint32_t vo = ((uint32_t)(2 * g) * lm_rows + row0 + (uint32_t)c) * 4u;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int kt = 0; kt < 2; kt++) {
                    const uint32_t lo = __builtin_amdgcn_raw_buffer_load_b32(rs_enc, (int)(vo + t * 64u), (int)((8u * kt) * plane), 0);
                    const uint32_t hi = __builtin_amdgcn_raw_buffer_load_b32(rs_enc, (int)(vo + t * 64u), (int)((8u * kt + 1u) * plane), 0);
                    xf_n[t][kt] = __builtin_bit_cast(h4, uint2{lo, hi});
                }
        } else {
            const uint32_t vo = (row0 + (uint32_t)c) * 64u + 8u * (uint32_t)g;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int kt = 0; kt < 2; kt++) {
                    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                    const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rs_enc, (int)(vo + t * 1024u + kt * 32u), 0, 0);
                    xf_n[t][kt] = __builtin_bit_cast(h4, v);
                }
        }
        if constexpr (COLOR) {
            const uint32_t vo = (row0 + (uint32_t)lane) * 12u;
#pragma unroll
            for (int k = 0; k < 3; k++) d_n[k] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_dir, (int)(vo + 4u * k), 0, 0));
        }
    };
    if (wave0 < n_groups) request(wave0);
    __syncthreads();                                                     // the image is complete
    [[maybe_unused]] int stamp_i = 1;
    for (uint32_t grp = wave0; grp < n_groups; grp += nwaves) {
        MLP_STAMP(blockIdx.x * WAVES + w, stamp_i); stamp_i++;
        h4 xf[NT][2];
#pragma unroll
        for (int t = 0; t < NT; t++) { xf[t][0] = xf_n[t][0]; xf[t][1] = xf_n[t][1]; }
        const float dx = d_n[0], dy = d_n[1], dz = d_n[2];
        if (grp + nwaves < n_groups) request(grp + nwaves);
        const uint32_t row_l = grp * 64u + (uint32_t)lane;
        if constexpr (COLOR) {
            float o[16], gx[1], gy[1], gz[1];
            sh_eval<4, false>(dx, dy, dz, o, gx, gy, gz);
            h8 lo, hi;
#pragma unroll
            for (int j = 0; j < 8; j++) { lo[j] = (half_t)o[j]; hi[j] = (half_t)o[8 + j]; }
            *reinterpret_cast<h8*>(&sc->sh[lane][0]) = lo;
            *reinterpret_cast<h8*>(&sc->sh[lane][8]) = hi;
        }
        h4 a0[NT][4], a1[NT][4];
        layer64s<1, NT>(img + I::S0, lane, xf, a0);
        layer64s<2, NT>(img + I::S1, lane, a0, a1);
        f4 so[NT];
        out16s<NT>(img + I::SO, lane, a1, so);
        h4 hq[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) hq[t][r] = (half_t)so[t][r];
            const uint32_t tile = grp * NT + t;
            if (h_out && tile < n_tiles) *reinterpret_cast<h4*>(h_out + ((size_t)tile * 16 + c) * 16 + 4 * g) = hq[t];
            if (g == 0) sc->q[t * 16 + c][0] = (float)hq[t][0];
        }
        wave_lds_fence();
        if (row_l < n_rows) sigmas[row_l] = density_scale * expf(sc->q[lane][0]);        // trunc_exp forward (activation.py:9)
        if constexpr (!COLOR) { wave_lds_fence(); continue; }
        h4 cin[NT][2];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            cin[t][0] = *reinterpret_cast<const h4*>(&sc->sh[t * 16 + c][4 * g]);
            const half_t nxt = __builtin_bit_cast(half_t, (uint16_t)__shfl_down((int)__builtin_bit_cast(uint16_t, hq[t][0]), 16, 64));
            cin[t][1][0] = hq[t][1]; cin[t][1][1] = hq[t][2]; cin[t][1][2] = hq[t][3];
            cin[t][1][3] = g == 3 ? (half_t)0.0f : nxt;
        }
        layer64s<1, NT>(img + I::C0, lane, cin, a0);
        layer64s<2, NT>(img + I::C1, lane, a0, a1);
        layer64s<2, NT>(img + I::C2, lane, a1, a0);
        f4 co[NT];
        out16s<NT>(img + I::CO, lane, a0, co);
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < NT; t++)
            if (g == 0) *reinterpret_cast<f4*>(&sc->q[t * 16 + c][0]) = co[t];
        wave_lds_fence();
        if (row_l < n_rows) {
            const f4 v = *reinterpret_cast<const f4*>(&sc->q[lane][0]);
            struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
            F3 out;
            out.x = (float)(half_t)(1.0f / (1.0f + expf(-(float)(half_t)v[0])));
            out.y = (float)(half_t)(1.0f / (1.0f + expf(-(float)(half_t)v[1])));
            out.z = (float)(half_t)(1.0f / (1.0f + expf(-(float)(half_t)v[2])));
            *reinterpret_cast<F3*>(rgbs + (size_t)row_l * 3) = out;
        }
        wave_lds_fence();
    }
    MLP_STAMP(blockIdx.x * WAVES + w, 15);
}

// ---------------------------------------------------------------- frame loop: head + composite_rays in one launch
// k_nerf_head_fwd5<true> followed, inside the wave, by composite_rays / composite_rays_distill (raymarching.cu:948-1035 /
// :1037-1142) for the rays whose rows the wave just shaded.  Rows are ray-major in 64-row groups of whole rays
// (lae_common.h FrameCtrl), lane L of a wave does the per-row work of row L of its group, so the sigma / rgb of a ray's
// n_step rows sit in n_step consecutive lanes: they go through the wave's LDS scratch to the lane of the ray's first row,
// which runs the reference's serial loop (same operations in the same order: the image is the same bits as the two-kernel
// form's) on the ray's 32-byte accumulator record.  sigmas / rgbs never reach memory (16 B written + 16 B read per row) and
// the compositing launch with its 40-56 B per ray of scattered accumulator traffic is gone: 9 / 29 us per iteration of the
// 800x800 / 1080p frame (profiles/r4_*).
// Work is dealt in CONTIGUOUS runs of groups, one run per wave (unit u = blockIdx.x * WAVES + wave), and a wave appends the
// rays that go on to its own survivor segment [u * R, ...) + count (+ the workgroup's sum of counts): the next iteration's
// per-ray kernels walk the segments in unit order (frame.hip frame_locate), so the alive list keeps its order -- stable
// compaction, no atomics, the same list every run.  (Compacting inside this launch -- every workgroup publishing its count
// and waiting for the counts of the workgroups before it -- was built and measured in round 4: 12.65 against 12.26 ms per
// 800x800 frame, and two processes sharing one GPU crawl when spinning workgroups keep each other's predecessors out.)
struct FrameHeadScratch { float dl[64][2]; uint32_t eo[64]; };         // per wave, behind its Head4Scratch

template <bool EDIT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_frame_head(
    const half_t* __restrict__ enc, const float* __restrict__ dirs, const half_t* __restrict__ Ws, const half_t* __restrict__ Wc,
    float density_scale, uint32_t lm_rows, lae::FrameHeadArgs fa) {
    constexpr int NT = 4, NTH = 64 * WAVES;
    using I = Head5Img;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t unit = blockIdx.x * WAVES + (uint32_t)w, n_units = gridDim.x * WAVES;
    const lae::FrameCtrl ctl = *fa.cur;
    const uint32_t n_alive = ctl.n_alive, n_step = max(ctl.n_step, 1u);
    const uint32_t rpg = lae::frame_rays_per_group(n_step);
    const uint32_t n_groups = (n_alive + rpg - 1) / rpg;
    const uint32_t per_unit = (n_groups + n_units - 1) / n_units;
    // the block's first unit decides whether the block has any work (units are dealt in order)
    if (min(n_groups, blockIdx.x * WAVES * per_unit) >= n_groups) {
        if (lane == 0) fa.seg_counts_next[unit] = 0u;
        if (threadIdx.x == 0) fa.blk_counts_next[blockIdx.x] = 0u;
        return;
    }
    __shared__ uint32_t wave_kept[WAVES];
    uint32_t kept = 0;                                                   // survivors this wave has appended (uniform)
    int32_t* seg = fa.seg_next + (size_t)unit * fa.R;
    const uint32_t g_lo = min(n_groups, unit * per_unit), g_hi = min(n_groups, g_lo + per_unit);
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* img = lds;
    Head4Scratch* sc = reinterpret_cast<Head4Scratch*>(lds + I::END) + w;
    FrameHeadScratch* fs = reinterpret_cast<FrameHeadScratch*>(reinterpret_cast<Head4Scratch*>(lds + I::END) + WAVES) + w;
    {   // weights -> swizzled LDS image (k_nerf_head_fwd5)
        constexpr int N1 = (4 * 1 * 64 + NTH - 1) / NTH, N2 = (4 * 2 * 64 + NTH - 1) / NTH, NO = (1 * 2 * 64 + NTH - 1) / NTH;
        uint4 b0[N1], b1[N2], b2[NO], c0[N1], c1[N2], c2[N2], c3[NO];
        stage_swizzled_issue<4, 1, NTH>(Ws, b0);
        stage_swizzled_issue<4, 2, NTH>(Ws + 64 * 32, b1);
        stage_swizzled_issue<1, 2, NTH>(Ws + 64 * 32 + 4096, b2);
        stage_swizzled_issue<4, 1, NTH>(Wc, c0);
        stage_swizzled_issue<4, 2, NTH>(Wc + 64 * 32, c1);
        stage_swizzled_issue<4, 2, NTH>(Wc + 64 * 32 + 4096, c2);
        stage_swizzled_issue<1, 2, NTH>(Wc + 64 * 32 + 8192, c3);
        stage_swizzled_store<4, 1, NTH>(img + I::C0, c0);
        stage_swizzled_store<4, 2, NTH>(img + I::C1, c1);
        stage_swizzled_store<4, 2, NTH>(img + I::C2, c2);
        stage_swizzled_store<1, 2, NTH>(img + I::CO, c3);
        stage_swizzled_store<4, 1, NTH>(img + I::S0, b0);
        stage_swizzled_store<4, 2, NTH>(img + I::S1, b1);
        stage_swizzled_store<1, 2, NTH>(img + I::SO, b2);
    }
    const uint32_t n_rows = (ctl.n_rows + 15u) & ~15u;                 // rows the emit kernel wrote (pad rows of the last tile are zeros)
    const __amdgpu_buffer_rsrc_t rs_enc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(enc), 0, (int)(lm_rows * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dir = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dirs), 0, (int)(n_rows * 12u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fa.deltas), 0, (int)(n_rows * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_eo = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(EDIT ? fa.edit_occ : reinterpret_cast<const uint8_t*>(fa.deltas)), 0, (int)n_rows, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_alive = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(fa.alive), 0, (int)(n_alive * 4u), 0x00020000);
    const uint32_t plane = lm_rows * 4u;
    // the lane of a ray's first row owns the ray
    const uint32_t slot = (uint32_t)lane / n_step;
    const bool own = slot * n_step == (uint32_t)lane && slot < rpg;
    h4 xf_n[NT][2] = {};
    float d_n[3] = {0.f, 0.f, 0.f}, dl_n[2] = {0.f, 0.f};
    uint32_t eo_n = 0, idx_n = 0;
    auto request = [&](uint32_t grp) {
        const uint32_t row0 = grp * 64u;
        const uint32_t vo = ((uint32_t)(2 * g) * lm_rows + row0 + (uint32_t)c) * 4u;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int kt = 0; kt < 2; kt++) {
                const uint32_t lo = __builtin_amdgcn_raw_buffer_load_b32(rs_enc, (int)(vo + t * 64u), (int)((8u * kt) * plane), 0);
                const uint32_t hi = __builtin_amdgcn_raw_buffer_load_b32(rs_enc, (int)(vo + t * 64u), (int)((8u * kt + 1u) * plane), 0);
                xf_n[t][kt] = __builtin_bit_cast(h4, uint2{lo, hi});
            }
        const uint32_t r = row0 + (uint32_t)lane;
#pragma unroll
        for (int k = 0; k < 3; k++) d_n[k] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_dir, (int)(r * 12u + 4u * k), 0, 0));
#pragma unroll
        for (int k = 0; k < 2; k++) dl_n[k] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_dl, (int)(r * 8u + 4u * k), 0, 0));
        if (EDIT) eo_n = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs_eo, (int)r, 0, 0);
        idx_n = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_alive, (int)((grp * rpg + slot) * 4u), 0, 0);   // 0 past the list
    };
    if (g_lo < g_hi) request(g_lo);
    __syncthreads();                                                     // the image is complete
    for (uint32_t grp = g_lo; grp < g_hi; grp++) {
        h4 xf[NT][2];
#pragma unroll
        for (int t = 0; t < NT; t++) { xf[t][0] = xf_n[t][0]; xf[t][1] = xf_n[t][1]; }
        const float dx = d_n[0], dy = d_n[1], dz = d_n[2], d0 = dl_n[0], d1 = dl_n[1];
        const uint32_t eo = eo_n, idx = idx_n;
        if (grp + 1 < g_hi) request(grp + 1);
        const bool valid = own && grp * rpg + slot < n_alive;
        // the ray's accumulators: requested now, needed after the two networks
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        lae::RayAcc* ap = fa.acc + idx;
        if (valid) { a0 = reinterpret_cast<const float4*>(ap)[0]; a1 = reinterpret_cast<const float4*>(ap)[1]; }
        {
            float o[16], gx[1], gy[1], gz[1];
            sh_eval<4, false>(dx, dy, dz, o, gx, gy, gz);
            h8 lo, hi;
#pragma unroll
            for (int j = 0; j < 8; j++) { lo[j] = (half_t)o[j]; hi[j] = (half_t)o[8 + j]; }
            *reinterpret_cast<h8*>(&sc->sh[lane][0]) = lo;
            *reinterpret_cast<h8*>(&sc->sh[lane][8]) = hi;
        }
        h4 a0h[NT][4], a1h[NT][4];
        layer64s<1, NT>(img + I::S0, lane, xf, a0h);
        layer64s<2, NT>(img + I::S1, lane, a0h, a1h);
        f4 so[NT];
        out16s<NT>(img + I::SO, lane, a1h, so);
        h4 hq[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) hq[t][r] = (half_t)so[t][r];
            if (g == 0) sc->q[t * 16 + c][0] = (float)hq[t][0];
        }
        wave_lds_fence();
        const float sig = density_scale * expf(sc->q[lane][0]);          // trunc_exp forward (activation.py:9), renderer.py:370
        h4 cin[NT][2];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            cin[t][0] = *reinterpret_cast<const h4*>(&sc->sh[t * 16 + c][4 * g]);
            const half_t nxt = __builtin_bit_cast(half_t, (uint16_t)__shfl_down((int)__builtin_bit_cast(uint16_t, hq[t][0]), 16, 64));
            cin[t][1][0] = hq[t][1]; cin[t][1][1] = hq[t][2]; cin[t][1][2] = hq[t][3];
            cin[t][1][3] = g == 3 ? (half_t)0.0f : nxt;
        }
        layer64s<1, NT>(img + I::C0, lane, cin, a0h);
        layer64s<2, NT>(img + I::C1, lane, a0h, a1h);
        layer64s<2, NT>(img + I::C2, lane, a1h, a0h);
        f4 co[NT];
        out16s<NT>(img + I::CO, lane, a0h, co);
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < NT; t++)
            if (g == 0) *reinterpret_cast<f4*>(&sc->q[t * 16 + c][0]) = co[t];
        wave_lds_fence();
        {
            const f4 v = *reinterpret_cast<const f4*>(&sc->q[lane][0]);
            const float cr = (float)(half_t)(1.0f / (1.0f + expf(-(float)(half_t)v[0])));
            const float cg = (float)(half_t)(1.0f / (1.0f + expf(-(float)(half_t)v[1])));
            const float cb = (float)(half_t)(1.0f / (1.0f + expf(-(float)(half_t)v[2])));
            wave_lds_fence();                                            // every lane has read its logits
            *reinterpret_cast<f4*>(&sc->q[lane][0]) = f4{sig, cr, cg, cb};
            fs->dl[lane][0] = d0; fs->dl[lane][1] = d1;
            if (EDIT) fs->eo[lane] = eo;
        }
        wave_lds_fence();
        // ---- composite_rays of the group's rays (raymarching.cu:948-1035; k_frame_composite of round 1-3, same statements)
        bool keep = false;
        if (valid) {
            float ws = a0.x, d = a0.y, r = a0.z, gc = a0.w, b = a1.x, t = a1.y, wse = a1.z, de = a1.w;
            uint32_t step = 0;
            while (step < n_step) {
                const float e0 = fs->dl[lane + step][0];
                if (e0 == 0) break;
                const f4 v = *reinterpret_cast<const f4*>(&sc->q[lane + step][0]);
                const float alpha = 1.0f - __expf(-v[0] * e0);
                const float T = 1 - ws;
                const float wgt = alpha * T;
                ws += wgt;
                if (EDIT) { if (fs->eo[lane + step]) { wse += wgt; de = fmaf(wgt, t, de); } }
                t += fs->dl[lane + step][1];
                d = fmaf(wgt, t, d);
                r = fmaf(wgt, v[1], r); gc = fmaf(wgt, v[2], gc); b = fmaf(wgt, v[3], b);
                if (T < fa.T_thresh) break;
                step++;
            }
            keep = step == n_step;
            reinterpret_cast<float4*>(ap)[0] = make_float4(ws, d, r, gc);
            reinterpret_cast<float4*>(ap)[1] = make_float4(b, keep ? t : a1.y, wse, de);      // rays_t moves only for rays that go on
        }
        const unsigned long long km = __ballot(keep);
        if (keep) seg[kept + (uint32_t)__builtin_popcountll(km & ((1ull << lane) - 1ull))] = (int32_t)idx;
        kept += (uint32_t)__builtin_popcountll(km);
        wave_lds_fence();                                                // scratch is rewritten by the next group
    }
    if (lane == 0) { fa.seg_counts_next[unit] = kept; wave_kept[w] = kept; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) tot += wave_kept[i];
        fa.blk_counts_next[blockIdx.x] = tot;
    }
}

// FFMLP forward for the common shape (64 wide, ReLU, linear output, 32 / 48 / 64 inputs, 1 or 2 hidden GEMMs) with the
// weights staged in LDS once per workgroup, like k_nerf_head_fwd: the generic k_mlp_fwd re-reads every weight fragment
// from global memory for every 32-row group (41 us per 100 k rows for the LAENeRF nets against ~10 us here).
template <int KT0, int NH>
struct Fwd64Cfg {
    static constexpr int IN = 16 * KT0, LDX = IN + 8, LDH = 72;
    static constexpr int W0 = 0, WH = W0 + 64 * LDX, WO = WH + NH * 64 * LDH, LDS_HALVES = WO + 16 * LDH;
};
template <int KT0, int NH>
__global__ __launch_bounds__(256) void k_mlp_fwd64(const half_t* __restrict__ in, const half_t* __restrict__ W, uint32_t n_tiles,
                                                   half_t* __restrict__ fwd_buf, half_t* __restrict__ out) {
    using C = Fwd64Cfg<KT0, NH>;
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    stage_rows(lds + C::W0, C::LDX, W, 64, C::IN);
#pragma unroll
    for (int l = 0; l < NH; l++) stage_rows(lds + C::WH + l * 64 * C::LDH, C::LDH, W + 64 * C::IN + (size_t)l * 4096, 64, 64);
    stage_rows(lds + C::WO, C::LDH, W + 64 * C::IN + (size_t)NH * 4096, 16, 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const uint32_t wave0 = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = gridDim.x * 4;
    const size_t B = (size_t)n_tiles * 16;
    for (uint32_t tile = wave0; tile < n_tiles; tile += nwaves) {
        const size_t row = (size_t)tile * 16 + c;
        h4 xf[KT0];
#pragma unroll
        for (int kt = 0; kt < KT0; kt++) xf[kt] = *reinterpret_cast<const h4*>(in + row * C::IN + kt * 16 + 4 * g);
        f4 acc[4];
        h4 a[4], b[4];
        layer64<KT0>(lds + C::W0, C::LDX, xf, c, g, acc); relu4(acc, a);
        if (fwd_buf) store_tiles<4>(fwd_buf, row, 64, g, a);
#pragma unroll
        for (int l = 0; l < NH; l++) {
            layer64<4>(lds + C::WH + l * 64 * C::LDH, C::LDH, a, c, g, acc); relu4(acc, b);
            if (fwd_buf) store_tiles<4>(fwd_buf + (size_t)(l + 1) * B * 64, row, 64, g, b);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) a[mt] = b[mt];
        }
        const f4 o = out16(lds + C::WO, C::LDH, a, c, g);
        h4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = (half_t)o[r];
        *reinterpret_cast<h4*>(out + row * 16 + 4 * g) = v;
    }
}

template <int KT0, int NH>
int launch_fwd64(const half_t* in, const half_t* W, uint32_t B, half_t* fwd_buf, half_t* out, hipStream_t s) {
    using C = Fwd64Cfg<KT0, NH>;
    const size_t lds_bytes = (size_t)C::LDS_HALVES * 2;
    const uint32_t n_tiles = B / 16;
    const uint32_t blocks = std::max(1u, std::min(lae::cdiv(n_tiles, 4), (uint32_t)lae::num_cus() * 2));
    k_mlp_fwd64<KT0, NH><<<blocks, 256, lds_bytes, s>>>(in, W, n_tiles, fwd_buf, out);
    return LAE_OK;
}

// 0 = fused backward where available (default), 1 = always the buffer-faithful three-kernel path
int g_ffmlp_mode = 0;

// ---------------------------------------------------------------- host side
int n_cus() { return lae::num_cus(); }

bool shape_ok(uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t hidden, uint32_t num_layers) {
    // ffmlp.py:112-115,157: hidden in {16..256}, in % 16 == 0, out <= 16 (always padded to 16), layers >= 2, B % 128 == 0
    const bool hid = hidden == 16 || hidden == 32 || hidden == 64 || hidden == 128 || hidden == 256;
    return hid && in_dim > 0 && in_dim % 16 == 0 && out_dim == 16 && num_layers >= 2 && B % 16 == 0;
}

template <int WIDTH, int NT>
void launch_fwd(const half_t* in, const half_t* W, uint32_t B, uint32_t in_dim, uint32_t n_hidden, uint32_t act,
                uint32_t out_act, half_t* fwd_buf, half_t* out, hipStream_t s) {
    const uint32_t n_groups = B / (16 * NT);
    const uint32_t blocks = min(lae::cdiv(n_groups, MLP_BLOCK / 64), (uint32_t)n_cus() * 4);
    k_mlp_fwd<WIDTH, NT><<<blocks, MLP_BLOCK, 0, s>>>(in, W, in_dim, n_hidden, act, out_act, fwd_buf, out, B, n_groups);
}

template <int WIDTH>
void forward_w(const half_t* in, const half_t* W, uint32_t B, uint32_t in_dim, uint32_t n_hidden, uint32_t act,
               uint32_t out_act, half_t* fwd_buf, half_t* out, hipStream_t s) {
    constexpr int NT = WIDTH <= 64 ? 2 : 1;
    if (B % (16 * NT) == 0) launch_fwd<WIDTH, NT>(in, W, B, in_dim, n_hidden, act, out_act, fwd_buf, out, s);
    else launch_fwd<WIDTH, 1>(in, W, B, in_dim, n_hidden, act, out_act, fwd_buf, out, s);
}

int forward_any(const void* inputs, const void* weights, uint32_t B, uint32_t in_dim, uint32_t out_dim, uint32_t hidden,
                uint32_t num_layers, uint32_t act, uint32_t out_act, void* fwd_buf, void* outputs, void* stream) {
    if (B == 0) return LAE_OK;
    if (!inputs || !weights || !outputs) return LAE_ENULL;
    if (!shape_ok(B, in_dim, out_dim, hidden, num_layers) || act > 6 || out_act > 6) return LAE_EINVAL;
    const half_t* in = (const half_t*)inputs; const half_t* W = (const half_t*)weights;
    half_t* fb = (half_t*)fwd_buf; half_t* out = (half_t*)outputs;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint32_t nh = num_layers - 1;
    if (hidden == 64 && act == LAE_ACT_RELU && out_act == 6 && (nh == 1 || nh == 2) && (in_dim == 32 || in_dim == 48 || in_dim == 64) &&
        g_ffmlp_mode != 1) {                                  // weights in LDS (< 48 KiB: no attribute needed)
        // these are exactly the shapes lae_ffmlp_backward serves with the recompute backward, which never reads forward_buffer
        // (nor does the reference's Python, ffmlp.py:31-35): in this mode the buffer is left untouched -- 256 / 384 bytes per row
        // that the drop-in step (reference wrappers: the buffer is always passed) would write for nothing
        fb = nullptr;
        if (in_dim == 32) nh == 1 ? launch_fwd64<2, 1>(in, W, B, fb, out, s) : launch_fwd64<2, 2>(in, W, B, fb, out, s);
        else if (in_dim == 48) nh == 1 ? launch_fwd64<3, 1>(in, W, B, fb, out, s) : launch_fwd64<3, 2>(in, W, B, fb, out, s);
        else nh == 1 ? launch_fwd64<4, 1>(in, W, B, fb, out, s) : launch_fwd64<4, 2>(in, W, B, fb, out, s);
        return lae::check_launch("ffmlp_forward");
    }
    switch (hidden) {
        case 16: forward_w<16>(in, W, B, in_dim, nh, act, out_act, fb, out, s); break;
        case 32: forward_w<32>(in, W, B, in_dim, nh, act, out_act, fb, out, s); break;
        case 64: forward_w<64>(in, W, B, in_dim, nh, act, out_act, fb, out, s); break;
        case 128: forward_w<128>(in, W, B, in_dim, nh, act, out_act, fb, out, s); break;
        case 256: forward_w<256>(in, W, B, in_dim, nh, act, out_act, fb, out, s); break;
        default: return LAE_EINVAL;       // ffmlp.cu:658
    }
    return lae::check_launch("ffmlp_forward");
}

template <int WIDTH>
int backward_w(const half_t* grad, const half_t* in, const half_t* W, const half_t* fwd_buf, uint32_t B, uint32_t in_dim,
               uint32_t n_hidden, uint32_t act, half_t* bwd_buf, half_t* grad_in, half_t* gw, hipStream_t s) {
    constexpr int NT = WIDTH <= 64 ? 2 : 1;
    {
        const bool nt2 = NT == 2 && B % 32 == 0;
        const uint32_t n_groups = B / (nt2 ? 32 : 16);
        const uint32_t blocks = min(lae::cdiv(n_groups, MLP_BLOCK / 64), (uint32_t)n_cus() * 4);
        if (nt2) k_mlp_bwd<WIDTH, NT><<<blocks, MLP_BLOCK, 0, s>>>(grad, W, fwd_buf, in_dim, n_hidden, act, bwd_buf, grad_in, B, n_groups);
        else k_mlp_bwd<WIDTH, 1><<<blocks, MLP_BLOCK, 0, s>>>(grad, W, fwd_buf, in_dim, n_hidden, act, bwd_buf, grad_in, B, n_groups);
    }
    const uint32_t nW = WIDTH * (in_dim + WIDTH * n_hidden + 16);
    constexpr uint32_t MT = WIDTH / 16, MB = (MT + 3) / 4;
    const uint32_t NB0 = (in_dim / 16 + 3) / 4;
    const uint32_t n_jobs = MB * NB0 + MB * MB * n_hidden + MB;
    // slices: aim for ~4 workgroups per CU in total, at least 64 rows per slice, bounded by the workspace
    uint32_t n_slices = max(1u, min(B / 64, (uint32_t)(n_cus() * 2) / n_jobs));
    float* g_ws = reinterpret_cast<float*>(lae::workspace(lae::WS_FFMLP_SLABS, (size_t)n_slices * nW * sizeof(float), s));
    if (!g_ws) return LAE_ELAUNCH;
    uint32_t rows_per_slice = lae::cdiv(B, n_slices);
    rows_per_slice = (rows_per_slice + 63) / 64 * 64;
    n_slices = lae::cdiv(B, rows_per_slice);
    k_mlp_dw<WIDTH><<<dim3(n_slices, n_jobs), MLP_BLOCK, 0, s>>>(grad, in, fwd_buf, bwd_buf, B, in_dim, n_hidden, g_ws, nW,
                                                                 rows_per_slice);
    k_dw_reduce<<<lae::cdiv(nW, 64), 64 * DWR_GROUPS, 0, s>>>(g_ws, n_slices, nW, gw);
    return LAE_OK;
}

}  // namespace

static uint32_t head_blocks_per_cu() {
    static int v = 0;
    if (v == 0) { const char* e = getenv("LAE_HEAD_BLOCKS_PER_CU"); v = e ? atoi(e) : 1; if (v < 1 || v > 4) v = 1; }
    return (uint32_t)v;
}
static int head5_waves() {
    static int v = 0;
    if (v == 0) { const char* e = getenv("LAE_HEAD5_WAVES"); v = e ? atoi(e) : 8; if (v != 4 && v != 8 && v != 16) v = 8; }
    return v;
}
template <bool COLOR, int WAVES>
static int launch_head_fwd5_w(const half_t* enc, const float* dirs, const half_t* Ws, const half_t* Wc, uint32_t n_tiles, uint32_t launch_tiles,
                              float density_scale, half_t* h_out, float* sigmas, float* rgbs, int level_major,
                              const uint32_t* n_rows_dev, uint32_t lm_rows, hipStream_t s) {
    const size_t lds_bytes = (size_t)(COLOR ? Head5Img::END : Head5Img::C0) * 2 + (size_t)WAVES * sizeof(Head4Scratch);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nerf_head_fwd5<COLOR, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess) return LAE_ELAUNCH;
        attr_set = true;
    }
    const uint32_t n_groups = lae::cdiv(launch_tiles, 4);
    const uint32_t blocks = std::max(1u, std::min(lae::cdiv(n_groups, WAVES), (uint32_t)lae::num_cus() * head_blocks_per_cu()));
    k_nerf_head_fwd5<COLOR, WAVES><<<blocks, 64 * WAVES, lds_bytes, s>>>(enc, dirs, Ws, Wc, n_tiles, density_scale, h_out, sigmas, rgbs,
                                                                        level_major, n_rows_dev, lm_rows);
    return LAE_OK;
}
template <bool COLOR>
static int launch_head_fwd5(const half_t* enc, const float* dirs, const half_t* Ws, const half_t* Wc, uint32_t n_tiles, uint32_t launch_tiles,
                            float density_scale, half_t* h_out, float* sigmas, float* rgbs, int level_major,
                            const uint32_t* n_rows_dev, uint32_t lm_rows, hipStream_t s) {
    switch (head5_waves()) {
        case 4: return launch_head_fwd5_w<COLOR, 4>(enc, dirs, Ws, Wc, n_tiles, launch_tiles, density_scale, h_out, sigmas, rgbs, level_major, n_rows_dev, lm_rows, s);
        case 16: return launch_head_fwd5_w<COLOR, 16>(enc, dirs, Ws, Wc, n_tiles, launch_tiles, density_scale, h_out, sigmas, rgbs, level_major, n_rows_dev, lm_rows, s);
        default: return launch_head_fwd5_w<COLOR, 8>(enc, dirs, Ws, Wc, n_tiles, launch_tiles, density_scale, h_out, sigmas, rgbs, level_major, n_rows_dev, lm_rows, s);
    }
}
// k_nerf_head_fwd5 stages the weights with 16-byte loads and addresses rows through buffer descriptors with 32-bit byte
// offsets (row * 64, lm_rows * 64, row * 12): both are preconditions of every entry point below
static int head_args_ok(const void* ws, const void* wc, uint64_t rows, const char* who) {
    if (((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(wc)) & 15) != 0) {
        lae::set_last_error_str((std::string(who) + ": the weight pointers must be 16-byte aligned").c_str());
        return LAE_EINVAL;
    }
    if (rows * 64ull > 0xffffffffull) {
        lae::set_last_error_str((std::string(who) + ": more than 2^26 - 1 rows per call (32-bit row offsets); split the batch").c_str());
        return LAE_EINVAL;
    }
    return LAE_OK;
}

// frame loop (frame.hip lae_render_frame): level-major features [16, M_cap, 2], loop state in device memory
uint32_t lae::frame_head_max_blocks() { return (uint32_t)lae::num_cus() * head_blocks_per_cu(); }

template <bool EDIT, int WAVES>
static int launch_frame_head(const half_t* enc, const float* dirs, const half_t* Ws, const half_t* Wc, uint32_t lm_rows, uint32_t blocks,
                             float density_scale, const lae::FrameHeadArgs& fa, hipStream_t s) {
    const size_t lds_bytes = (size_t)Head5Img::END * 2 + (size_t)WAVES * (sizeof(Head4Scratch) + sizeof(FrameHeadScratch));
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_head<EDIT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess) return LAE_ELAUNCH;
        attr_set = true;
    }
    k_frame_head<EDIT, WAVES><<<blocks, 64 * WAVES, lds_bytes, s>>>(enc, dirs, Ws, Wc, density_scale, lm_rows, fa);
    return LAE_OK;
}

int lae::nerf_head_composite_frame(const void* enc, const float* dirs, const void* sigma_weights, const void* color_weights,
                                   uint32_t M_cap, float density_scale, const FrameHeadArgs& fa, bool edit, uint32_t n_blocks,
                                   hipStream_t stream) {
    if (M_cap % 64 != 0 || n_blocks == 0) return LAE_EINVAL;
    if (const int rc = head_args_ok(sigma_weights, color_weights, M_cap, "render_frame(head)")) return rc;
    const half_t *e = (const half_t*)enc, *ws = (const half_t*)sigma_weights, *wc = (const half_t*)color_weights;
    return edit ? launch_frame_head<true, FRAME_HEAD_WAVES>(e, dirs, ws, wc, M_cap, n_blocks, density_scale, fa, stream)
                : launch_frame_head<false, FRAME_HEAD_WAVES>(e, dirs, ws, wc, M_cap, n_blocks, density_scale, fa, stream);
}

extern "C" {

int lae_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                      uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                      void* forward_buffer, void* outputs, void* stream) {
    // forward_buffer == NULL is accepted: activations are then not saved (the fused backward recomputes them)
    return forward_any(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                       forward_buffer, outputs, stream);
}

int lae_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                        uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                        void* inference_buffer, void* outputs, void* stream) {
    (void)inference_buffer;   // activations never leave the registers
    return forward_any(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                       nullptr, outputs, stream);
}

int lae_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B,
                       uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                       uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,
                       void* grad_inputs, void* grad_weights, void* stream) {
    return lae_ffmlp_backward_ex(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                                 output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, 0, nullptr, stream);
}

int lae_ffmlp_backward_ex(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B,
                          uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                          uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,
                          void* grad_inputs, void* grad_weights, int accumulate, int32_t* nonfinite_flag, void* stream) {
    (void)output_activation;   // the reference ignores it too (ffmlp.cu:781)
    if (B == 0) return LAE_OK;
    if (!grad || !inputs || !weights || !grad_weights) return LAE_ENULL;
    if (calc_grad_inputs && !grad_inputs) return LAE_ENULL;
    if (!shape_ok(B, input_dim, output_dim, hidden_dim, num_layers) || activation > 6) return LAE_EINVAL;
    const half_t* g = (const half_t*)grad; const half_t* in = (const half_t*)inputs; const half_t* W = (const half_t*)weights;
    const half_t* fb = (const half_t*)forward_buffer; half_t* bb = (half_t*)backward_buffer;
    half_t* gi = calc_grad_inputs ? (half_t*)grad_inputs : nullptr; half_t* gw = (half_t*)grad_weights;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint32_t nh = num_layers - 1;
    int rc;
    // fused path: recomputes the activations, never touches forward_buffer / backward_buffer (both are scratch
    // that the reference's Python allocates and never reads, ffmlp.py:31-35,71)
    if (g_ffmlp_mode != 1 && hidden_dim == 64 && activation == LAE_ACT_RELU && (nh == 1 || nh == 2) &&
        (input_dim == 32 || input_dim == 48 || input_dim == 64)) {
        rc = LAE_EINVAL;
        if (nh == 1 && input_dim == 32) rc = launch_bwd_fused<32, 1>(g, in, W, B, gi, gw, s, HeadBwdArgs{}, accumulate, nullptr, nullptr, nonfinite_flag);
        else if (nh == 2 && input_dim == 32) rc = launch_bwd_fused<32, 2>(g, in, W, B, gi, gw, s, HeadBwdArgs{}, accumulate, nullptr, nullptr, nonfinite_flag);
        else if (nh == 1 && input_dim == 48) rc = launch_bwd_fused<48, 1>(g, in, W, B, gi, gw, s, HeadBwdArgs{}, accumulate, nullptr, nullptr, nonfinite_flag);
        else if (nh == 2 && input_dim == 48) rc = launch_bwd_fused<48, 2>(g, in, W, B, gi, gw, s, HeadBwdArgs{}, accumulate, nullptr, nullptr, nonfinite_flag);
        else if (nh == 1 && input_dim == 64) rc = launch_bwd_fused<64, 1>(g, in, W, B, gi, gw, s, HeadBwdArgs{}, accumulate, nullptr, nullptr, nonfinite_flag);
        else if (nh == 2 && input_dim == 64) rc = launch_bwd_fused<64, 2>(g, in, W, B, gi, gw, s, HeadBwdArgs{}, accumulate, nullptr, nullptr, nonfinite_flag);
        if (rc) return rc;
        return lae::check_launch("ffmlp_backward(fused)");
    }
    if (accumulate || nonfinite_flag) {
        lae::set_last_error_str("ffmlp_backward_ex: accumulate / nonfinite_flag are served by the fused backward only (hidden 64, ReLU, 1-2 hidden GEMMs, input 32 / 48 / 64)");
        return LAE_EINVAL;
    }
    if (!forward_buffer || !backward_buffer) return LAE_ENULL;
    switch (hidden_dim) {
        case 16: rc = backward_w<16>(g, in, W, fb, B, input_dim, nh, activation, bb, gi, gw, s); break;
        case 32: rc = backward_w<32>(g, in, W, fb, B, input_dim, nh, activation, bb, gi, gw, s); break;
        case 64: rc = backward_w<64>(g, in, W, fb, B, input_dim, nh, activation, bb, gi, gw, s); break;
        case 128: rc = backward_w<128>(g, in, W, fb, B, input_dim, nh, activation, bb, gi, gw, s); break;
        case 256: rc = backward_w<256>(g, in, W, fb, B, input_dim, nh, activation, bb, gi, gw, s); break;
        default: return LAE_EINVAL;
    }
    if (rc) return rc;
    return lae::check_launch("ffmlp_backward");
}

int lae_nerf_head_forward(const void* enc, const float* dirs, const void* sigma_weights, const void* color_weights,
                          uint32_t M, float density_scale, void* h_out, float* sigmas, float* rgbs, int enc_level_major,
                          void* stream) {
    if (M == 0) return LAE_OK;
    if (!enc || !dirs || !sigma_weights || !color_weights || !h_out || !sigmas || !rgbs) return LAE_ENULL;
    if (M % 16 != 0) return LAE_EINVAL;
    if (const int rc = head_args_ok(sigma_weights, color_weights, M, "nerf_head_forward")) return rc;
    const int rc = launch_head_fwd5<true>((const half_t*)enc, dirs, (const half_t*)sigma_weights, (const half_t*)color_weights, M / 16, M / 16,
                                          density_scale, (half_t*)h_out, sigmas, rgbs, enc_level_major, nullptr, M,
                                          reinterpret_cast<hipStream_t>(stream));
    return rc != LAE_OK ? rc : lae::check_launch("nerf_head_forward");
}

int lae_nerf_density_forward(const void* enc, const void* sigma_weights, uint32_t M, float density_scale, void* h_out,
                             float* sigmas, int enc_level_major, void* stream) {
    if (M == 0) return LAE_OK;
    if (!enc || !sigma_weights || !sigmas) return LAE_ENULL;
    if (M % 16 != 0) return LAE_EINVAL;
    if (const int rc = head_args_ok(sigma_weights, sigma_weights, M, "nerf_density_forward")) return rc;
    const int rc = launch_head_fwd5<false>((const half_t*)enc, nullptr, (const half_t*)sigma_weights, nullptr, M / 16, M / 16, density_scale,
                                           (half_t*)h_out, sigmas, nullptr, enc_level_major, nullptr, M, reinterpret_cast<hipStream_t>(stream));
    return rc != LAE_OK ? rc : lae::check_launch("nerf_density_forward");
}

int lae_nerf_head_backward(const float* grad_sigmas, const float* grad_rgbs, const void* enc, const float* dirs, const void* h,
                           const float* rgbs, const void* sigma_weights, const void* color_weights, uint32_t M,
                           float density_scale, void* grad_h, void* grad_enc, void* grad_sigma_weights,
                           void* grad_color_weights, int accumulate_weight_grads, int enc_level_major, int32_t* nonfinite_flag,
                           const float* loss_partials, uint32_t loss_n_part, uint32_t loss_n_elem, const float* loss_scale,
                           float* loss_out, void* stream) {
    if (!grad_sigma_weights || !grad_color_weights) return LAE_ENULL;
    if (loss_out && (!loss_partials || loss_n_elem == 0)) return LAE_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (M == 0) {                                        // no samples: zero weight gradients (like the reference's GEMMs on empty batches)
        if (loss_out) {
            const int rc = lae_loss_finish(loss_partials, loss_n_part, loss_n_elem, loss_scale, loss_out, stream);
            if (rc != LAE_OK) return rc;
        }
        if (accumulate_weight_grads) return LAE_OK;
        if (hipMemsetAsync(grad_sigma_weights, 0, 64 * (32 + 64 + 16) * 2, s) != hipSuccess) return LAE_ELAUNCH;
        if (hipMemsetAsync(grad_color_weights, 0, 64 * (32 + 128 + 16) * 2, s) != hipSuccess) return LAE_ELAUNCH;
        return LAE_OK;
    }
    if (!grad_sigmas || !grad_rgbs || !enc || !dirs || !h || !rgbs || !sigma_weights || !color_weights || !grad_h) return LAE_ENULL;
    if (M % 16 != 0) return LAE_EINVAL;
    HeadBwdArgs ha{dirs, rgbs, grad_rgbs, grad_sigmas, density_scale, 0};
    HeadBwdArgs hs{};
    hs.level_major = enc_level_major;
    // partial slabs of both networks side by side in the workspace, reduced by ONE launch after the second backward kernel
    const uint32_t nW_c = 64 * (32 + 128 + 16), nW_s = 64 * (32 + 64 + 16);
    const size_t cap = (size_t)lae::num_cus() * 2;
    float* ws = reinterpret_cast<float*>(lae::workspace(lae::WS_FFMLP_SLABS, cap * (nW_c + nW_s) * sizeof(float), s));
    if (!ws) return LAE_ELAUNCH;
    float* ws_s = ws + cap * nW_c;
    uint32_t n_c = 0, n_s = 0;
    int rc = launch_bwd_fused<32, 2, 1>(nullptr, (const half_t*)h, (const half_t*)color_weights, M, (half_t*)grad_h,
                                        (half_t*)grad_color_weights, s, ha, accumulate_weight_grads, ws, &n_c);
    if (rc != LAE_OK) return rc;
    rc = launch_bwd_fused<32, 1, 0>((const half_t*)grad_h, (const half_t*)enc, (const half_t*)sigma_weights, M, (half_t*)grad_enc,
                                    (half_t*)grad_sigma_weights, s, hs, accumulate_weight_grads, ws_s, &n_s);
    if (rc != LAE_OK) return rc;
    const uint32_t nb_c = lae::cdiv(nW_c, 64u), nb_s = lae::cdiv(nW_s, 64u);
    const LossFinish lf{loss_partials, loss_n_part, loss_n_elem, loss_scale, loss_out};
    k_dw_reduce2<<<nb_c + nb_s + (loss_out ? 1u : 0u), 64 * DWR_GROUPS, 0, s>>>(ws, n_c, nW_c, (half_t*)grad_color_weights, ws_s, n_s, nW_s,
                                                                               (half_t*)grad_sigma_weights, nb_c, nb_s,
                                                                               accumulate_weight_grads, nonfinite_flag, lf);
    return lae::check_launch("nerf_head_backward");
}

int lae_ffmlp_set_mode(int mode) {
    // 0: default (fused backward, wave-private dW with MFMA transposes); 1: buffer-faithful three-kernel backward (the reference's
    // forward / backward buffers); 3: fused backward with the round-2 workgroup-cooperative dW kernel (the one A/B predecessor)
    if (mode != 0 && mode != 1 && mode != 3) return LAE_EINVAL;
    g_ffmlp_mode = mode == 1 ? 1 : 0;
    g_bwd_fused_variant = mode == 3 ? 0 : 2;
    return LAE_OK;
}

int lae_allocate_splitk(uint64_t size) {
    (void)size;   // number of side streams in the reference; one stream suffices here.
    return LAE_OK;  // the slab workspace is allocated on the first backward call (needs a device)
}

int lae_free_splitk(void) {
    lae::free_workspaces();
    return LAE_OK;
}

}  // extern "C"
