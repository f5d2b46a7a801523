// loss.hip -- the training criterion of the reference's Trainer fused with the loss scaling.
//
// nerf/utils.py:560-640 (train_step): loss = MSELoss(reduction='none')(pred_rgb, gt_rgb).mean(-1).mean(), then
// GradScaler.scale(loss).backward().  In torch that is ~8 tiny kernels (sub/pow, mean, scale mul, ones_like, mul
// backward, mse backward, ...), each ~4 us inside a captured graph.  Here: ONE single-workgroup kernel produces
// the loss, the scaled loss and d(scaled loss)/d(pred); the backward is a multiplication by the incoming gradient.
#include "lae_common.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

// n elements (<= a few hundred thousand: one 1024-thread workgroup walks them; a 4096-ray batch is 12288 elements)
__global__ __launch_bounds__(1024) void k_mse_fwd(const float* __restrict__ pred, const float* __restrict__ target, uint32_t n,
                                                  const float* __restrict__ scale, float* __restrict__ loss_out,
                                                  float* __restrict__ grad) {
    __shared__ float part[16];
    const float s = scale ? scale[0] : 1.0f;
    const float gk = 2.0f / (float)n;
    float acc = 0.0f;
    // rounds of 4 elements per thread, every load of a round in flight before the first is used (one workgroup: the
    // kernel's time is its chain of memory latencies, 12 of them with one element per thread and round)
    constexpr uint32_t PER = 4;
    for (uint32_t base = 0; base < n; base += 1024 * PER) {
        float p[PER], t[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = base + k * 1024 + threadIdx.x;
            p[k] = i < n ? pred[i] : 0.0f; t[k] = i < n ? target[i] : 0.0f;
        }
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = base + k * 1024 + threadIdx.x;
            const float d = p[k] - t[k];
            acc = fmaf(d, d, acc);
            if (i < n) grad[i] = (d * gk) * s;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; w++) t += part[w];
        const float loss = t / (float)n;
        loss_out[0] = loss * s;                             // what backward() is called on
        loss_out[1] = loss;                                 // for logging
    }
}

}  // namespace

extern "C" {

int lae_mse_loss_forward(const float* pred, const float* target, uint32_t n, const float* scale, float* loss_out, float* grad,
                         void* stream) {
    if (!pred || !target || !loss_out || !grad) return LAE_ENULL;
    if (n == 0) return LAE_EINVAL;
    k_mse_fwd<<<1, 1024, 0, STREAM(stream)>>>(pred, target, n, scale, loss_out, grad);
    return lae::check_launch("mse_loss_forward");
}

}  // extern "C"
