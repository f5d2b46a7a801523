// shencoder.hip -- real spherical-harmonics direction encoder for gfx950.
//
// Replaces shencoder/src/shencoder.cu of the reference (kernel_sh :27-355,
// kernel_sh_backward :358-382).  The reference spells out 64 + 3*64 expanded
// polynomials; here the basis is evaluated in factored form
//     Y[l,+m] = Q_lm(z) * Re(x+iy)^m ,  Y[l,-m] = Q_lm(z) * Im(x+iy)^m
// from a generated coefficient table (tools/gen_sh_table.py -> sh_table.inc), with
// the same index order and sign convention.  One lane = one direction; the row of
// degree^2 outputs is staged through LDS so the wave writes full 256-byte lines.
#include "lae_common.h"

namespace {

#include "sh_table.inc"

// degree 8 needs 65 floats of LDS per lane: 128 lanes keep the tile under the 64 KiB static limit
constexpr int sh_block(int deg) { return deg >= 8 ? 128 : 256; }

template <int DEG, bool GRAD>
__global__ __launch_bounds__(sh_block(DEG)) void k_sh_fwd(const float* __restrict__ inputs, float* __restrict__ outputs,
                                                      uint32_t B, float* __restrict__ dy_dx) {
    constexpr int C2 = DEG * DEG;
    constexpr int SH_BLOCK = sh_block(DEG);
    const uint32_t b = blockIdx.x * SH_BLOCK + threadIdx.x;
    const bool live = b < B;
    float x = 0, y = 0, z = 0;
    if (live) { x = inputs[3 * (size_t)b]; y = inputs[3 * (size_t)b + 1]; z = inputs[3 * (size_t)b + 2]; }
    float o[C2];
    float gx[GRAD ? C2 : 1], gy[GRAD ? C2 : 1], gz[GRAD ? C2 : 1];
    sh_eval<DEG, GRAD>(x, y, z, o, gx, gy, gz);

    // transpose through LDS: lane-major rows -> block-contiguous output (C2 floats per sample).
    // +1 padding keeps the lane-strided writes conflict free.
    __shared__ float tile[SH_BLOCK * (C2 + 1)];
    const uint32_t base = blockIdx.x * SH_BLOCK;
    const uint32_t nrow = min((uint32_t)SH_BLOCK, B - base);
    auto flush = [&](const float (&v)[C2], float* dst, uint32_t row_stride, uint32_t col_off) {
#pragma unroll
        for (int i = 0; i < C2; i++) tile[threadIdx.x * (C2 + 1) + i] = v[i];
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < nrow * C2; e += SH_BLOCK) {
            const uint32_t r = e / C2, c = e - r * C2;
            dst[(size_t)(base + r) * row_stride + col_off + c] = tile[r * (C2 + 1) + c];
        }
        __syncthreads();
    };
    flush(o, outputs, C2, 0);
    if constexpr (GRAD) {       // dy_dx[b] = [dx block | dy block | dz block], shencoder.cu:125-128
        flush(gx, dy_dx, 3 * C2, 0);
        flush(gy, dy_dx, 3 * C2, C2);
        flush(gz, dy_dx, 3 * C2, 2 * C2);
    }
}

// shencoder.cu:358-382: grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]
__global__ void k_sh_bwd(const float* __restrict__ grad, uint32_t B, uint32_t C2, const float* __restrict__ dy_dx,
                         float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / 3;
    if (b >= B) return;
    const uint32_t d = t - b * 3;
    const float* g = grad + (size_t)b * C2;
    const float* dd = dy_dx + (size_t)b * 3 * C2 + (size_t)d * C2;
    float r = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) r = fmaf(g[ch], dd[ch], r);     // shencoder.cu:378 `+=` of a product
    grad_inputs[t] = r;
}

template <int DEG>
static void launch_sh(const float* in, float* out, uint32_t B, float* dy_dx, hipStream_t s) {
    constexpr int SH_BLOCK = sh_block(DEG);
    const uint32_t nb = lae::cdiv(B, SH_BLOCK);
    if (dy_dx) k_sh_fwd<DEG, true><<<nb, SH_BLOCK, 0, s>>>(in, out, B, dy_dx);
    else k_sh_fwd<DEG, false><<<nb, SH_BLOCK, 0, s>>>(in, out, B, nullptr);
}

}  // namespace

extern "C" {

int lae_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx,
                          void* stream) {
    if (B == 0) return LAE_OK;
    if (!inputs || !outputs) return LAE_ENULL;
    if (D != 3) return LAE_EINVAL;            // sphere_harmonics.py:69
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (C) {                              // sphere_harmonics.py:70 degree in [1, 8]
        case 1: launch_sh<1>(inputs, outputs, B, dy_dx, s); break;
        case 2: launch_sh<2>(inputs, outputs, B, dy_dx, s); break;
        case 3: launch_sh<3>(inputs, outputs, B, dy_dx, s); break;
        case 4: launch_sh<4>(inputs, outputs, B, dy_dx, s); break;
        case 5: launch_sh<5>(inputs, outputs, B, dy_dx, s); break;
        case 6: launch_sh<6>(inputs, outputs, B, dy_dx, s); break;
        case 7: launch_sh<7>(inputs, outputs, B, dy_dx, s); break;
        case 8: launch_sh<8>(inputs, outputs, B, dy_dx, s); break;
        default: return LAE_EINVAL;
    }
    return lae::check_launch("sh_encode_forward");
}

int lae_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                           const float* dy_dx, float* grad_inputs, void* stream) {
    (void)inputs;
    if (B == 0) return LAE_OK;
    if (!grad || !dy_dx || !grad_inputs) return LAE_ENULL;
    if (D != 3 || C < 1 || C > 8) return LAE_EINVAL;
    k_sh_bwd<<<lae::cdiv((uint64_t)B * 3, 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(grad, B, C * C, dy_dx,
                                                                                                  grad_inputs);
    return lae::check_launch("sh_encode_backward");
}

}  // extern "C"
