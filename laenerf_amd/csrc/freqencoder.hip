// freqencoder.hip -- NeRF frequency (sin/cos) encoder for gfx950.
//
// Replaces freqencoder/src/freqencoder.cu of the reference (K18/K19; only reachable through
// get_encoder('frequency'), encoding.py:59-62 -- no shipped config uses it).
//
// Streaming kernels: one lane per output element (consecutive lanes write consecutive floats), the backward one lane
// per (sample, input dim).  4*(D + C) bytes per sample forward; bound by HBM.
#include "lae_common.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

// freqencoder.cu:30-58
__global__ __launch_bounds__(256) void k_freq_fwd(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t C,
                                                  float* __restrict__ outputs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)B * C) return;
    const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
    const float* in = inputs + (size_t)b * D;
    if (c < D) { outputs[t] = in[c]; return; }
    const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
    const float phase = (float)(col % 2) * (3.141592653589793f / 2);
    outputs[t] = __sinf(scalbnf(in[d], (int)freq) + phase);          // the reference uses the fast sine too (:57)
}

// freqencoder.cu:63-94
__global__ __launch_bounds__(256) void k_freq_bwd(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B,
                                                  uint32_t D, uint32_t deg, uint32_t C, float* __restrict__ grad_inputs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)B * D) return;
    const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
    const float* g = grad + (size_t)b * C;
    const float* o = outputs + (size_t)b * C;
    float r = g[d];
    g += D; o += D;
    for (uint32_t f = 0; f < deg; f++) {
        r += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
        g += 2 * D; o += 2 * D;
    }
    grad_inputs[t] = r;
}

}  // namespace

extern "C" {

int lae_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs, void* stream) {
    if (B == 0) return LAE_OK;
    if (!inputs || !outputs) return LAE_ENULL;
    if (D == 0 || C != D + 2 * D * deg) return LAE_EINVAL;
    k_freq_fwd<<<lae::cdiv((uint64_t)B * C, 256), 256, 0, STREAM(stream)>>>(inputs, B, D, C, outputs);
    return lae::check_launch("freq_encode_forward");
}

int lae_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                             float* grad_inputs, void* stream) {
    if (B == 0) return LAE_OK;
    if (!grad || !outputs || !grad_inputs) return LAE_ENULL;
    if (D == 0 || C != D + 2 * D * deg) return LAE_EINVAL;
    k_freq_bwd<<<lae::cdiv((uint64_t)B * D, 256), 256, 0, STREAM(stream)>>>(grad, outputs, B, D, deg, C, grad_inputs);
    return lae::check_launch("freq_encode_backward");
}

}  // extern "C"
