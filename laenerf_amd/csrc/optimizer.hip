// optimizer.hip -- fused Adam + GradScaler for the NeRF parameters on gfx950.
//
// The reference trains with torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15) under torch.cuda.amp.GradScaler
// (main_nerf.py:223, nerf/utils.py:1474-1482).  Around the 12.2 M-parameter hash table that is, per step: a 25 MB
// zero fill of the fp16 gradient, a fp16->fp32 gradient cast, an unscale pass, the multi-tensor Adam passes and a
// fp32->fp16 cast of the table for the next forward (~575 MB of HBM traffic in ~20 launches).  Here:
//   k_check   any non-finite gradient?                                  (reads the fp16 gradient once)
//   k_begin   one thread: skip/step decision, bias corrections in double, scale growth/backoff (GradScaler.update)
//   k_apply   one pass: unscale, Adam, write p/m/v, write the fp16 shadow table the encoder gathers from,
//             zero the gradient buffer for the next accumulation       (30 B per parameter)
// All state lives on the device (OptState), so the whole step is HIP-graph capturable.
#include "lae_common.h"
#include <hip/hip_fp16.h>

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

typedef _Float16 half_t;

struct OptState {          // 16 words, see include/laenerf.h
    float scale;           // 0  current loss scale
    int32_t growth_tracker;// 1  consecutive finite steps
    int32_t found_inf;     // 2  accumulator written by k_check
    int32_t skip;          // 3  decision for k_apply
    int32_t step;          // 4  number of optimizer steps taken
    float inv_bc1;         // 5  1 / (1 - beta1^step)
    float bc2_sqrt;        // 6  sqrt(1 - beta2^step)
    float inv_scale;       // 7  1 / scale used by this step's gradients
    int32_t skipped_total; // 8  number of skipped steps (diagnostics)
    // round 3: the bias corrections of the NEXT step, left by the update launch (k_apply*) so that k_begin -- a one-thread
    // launch on the step's critical path -- needs no pow / sqrt / divide in double (it was ~5 us, most of it that arithmetic)
    float next_inv_bc1;    // 9   1 / (1 - beta1^next_for)
    float next_bc2_sqrt;   // 10  sqrt(1 - beta2^next_for)
    int32_t next_for;      // 11  the step count words 9 / 10 are for (0 = none: k_begin computes them itself)
    float next_beta1;      // 12  the betas they were computed with
    float next_beta2;      // 13
    int32_t pad[2];
};

__device__ __forceinline__ void bias_corrections(float beta1, float beta2, int step, float& inv_bc1, float& bc2_sqrt) {
    inv_bc1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
}
// called by ONE thread of the update launch, which only READS words 3-7: the values for the step after the one being applied
__device__ __forceinline__ void prepare_next_step(OptState* st, float beta1, float beta2) {
    const int nxt = st->step + 1;
    float a, b;
    bias_corrections(beta1, beta2, nxt, a, b);
    st->next_inv_bc1 = a; st->next_bc2_sqrt = b; st->next_beta1 = beta1; st->next_beta2 = beta2;
    st->next_for = nxt;
}

template <typename G> __device__ __forceinline__ float ldg(const G* g, size_t i) { return (float)g[i]; }

template <typename G>
__global__ __launch_bounds__(256) void k_check(const G* __restrict__ grad, size_t n, OptState* __restrict__ st) {
    bool bad = false;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            if constexpr (sizeof(G) == 2) {
                const uint4 v = *reinterpret_cast<const uint4*>(grad + i);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) bad |= ((w[k] & 0x7c00u) == 0x7c00u) || ((w[k] & 0x7c000000u) == 0x7c000000u);   // exponent all ones
            } else {
                const float4 a = *reinterpret_cast<const float4*>(grad + i), b = *reinterpret_cast<const float4*>(grad + i + 4);
                bad |= !isfinite(a.x) || !isfinite(a.y) || !isfinite(a.z) || !isfinite(a.w) || !isfinite(b.x) || !isfinite(b.y) ||
                       !isfinite(b.z) || !isfinite(b.w);
            }
        } else {
            for (size_t j = i; j < n; j++) bad |= !isfinite(ldg(grad, j));
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&st->found_inf, 1);
}

// GradScaler.step + update (torch/amp/grad_scaler.py) and the scalar part of Adam (torch/optim/adam.py, non-capturable
// formulas: bias corrections from beta ** step evaluated in double)
__global__ void k_begin(OptState* __restrict__ st, float beta1, float beta2, int growth_interval, float growth, float backoff,
                        int use_scaler) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool bad = use_scaler && st->found_inf != 0;
    st->found_inf = 0;
    st->skip = bad;
    st->inv_scale = use_scaler ? (float)(1.0 / (double)st->scale) : 1.0f;
    if (bad) {
        st->scale = st->scale * backoff;
        st->growth_tracker = 0;
        st->skipped_total += 1;
        return;
    }
    st->step += 1;
    if (st->next_for == st->step && st->next_beta1 == beta1 && st->next_beta2 == beta2) {      // left by the previous update launch
        st->inv_bc1 = st->next_inv_bc1; st->bc2_sqrt = st->next_bc2_sqrt;
    } else {                                                                                   // first step, loaded checkpoint, ...
        float a, b;
        bias_corrections(beta1, beta2, st->step, a, b);
        st->inv_bc1 = a; st->bc2_sqrt = b;
    }
    if (use_scaler) {
        st->growth_tracker += 1;
        if (st->growth_tracker == growth_interval) { st->scale = st->scale * growth; st->growth_tracker = 0; }
    }
}

struct AdamHyper { float beta1, beta2, eps, weight_decay; };

__device__ __forceinline__ void adam1(float& p, float& m, float& v, float g, float lr_over_bc1, float bc2_sqrt, const AdamHyper& h) {
    if (h.weight_decay != 0.0f) g = g + h.weight_decay * p;                 // adam.py: grad.add(param, alpha=weight_decay)
    m = m + (g - m) * (1.0f - h.beta1);                                     // exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2 + (1.0f - h.beta2) * (g * g);                           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / bc2_sqrt + h.eps;                        // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - lr_over_bc1 * (m / denom);                                      // param.addcdiv_(exp_avg, denom, value=-step_size)
}

template <typename G>
__global__ __launch_bounds__(256) void k_apply(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                               G* __restrict__ grad, half_t* __restrict__ shadow, size_t n,
                                               OptState* __restrict__ st, const float* __restrict__ lr_ptr, AdamHyper h) {
    const bool skip = st->skip != 0;
    const float inv_scale = st->inv_scale, bc2_sqrt = st->bc2_sqrt;
    const float lr_over_bc1 = (float)((double)lr_ptr[0] * (double)st->inv_bc1);     // step_size = lr / bias_correction1
    if (blockIdx.x == 0 && threadIdx.x == 0) prepare_next_step(st, h.beta1, h.beta2);
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float g[4];
            if constexpr (sizeof(G) == 2) {
                const uint2 raw = *reinterpret_cast<const uint2*>(grad + i);
                const half_t* gh = reinterpret_cast<const half_t*>(&raw);
#pragma unroll
                for (int k = 0; k < 4; k++) g[k] = (float)gh[k];
                *reinterpret_cast<uint2*>(grad + i) = uint2{0u, 0u};
            } else {
                const float4 raw = *reinterpret_cast<const float4*>(grad + i);
                g[0] = raw.x; g[1] = raw.y; g[2] = raw.z; g[3] = raw.w;
                *reinterpret_cast<float4*>(grad + i) = float4{0, 0, 0, 0};
            }
            if (skip) continue;
            float4 pp = *reinterpret_cast<const float4*>(p + i), mm = *reinterpret_cast<const float4*>(m + i),
                   vv = *reinterpret_cast<const float4*>(v + i);
            adam1(pp.x, mm.x, vv.x, g[0] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            adam1(pp.y, mm.y, vv.y, g[1] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            adam1(pp.z, mm.z, vv.z, g[2] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            adam1(pp.w, mm.w, vv.w, g[3] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
            if (shadow) {
                half_t s[4] = {(half_t)pp.x, (half_t)pp.y, (half_t)pp.z, (half_t)pp.w};
                *reinterpret_cast<uint2*>(shadow + i) = *reinterpret_cast<const uint2*>(s);
            }
        } else {
            for (size_t j = i; j < n; j++) {
                const float gj = ldg(grad, j) * inv_scale;
                grad[j] = (G)0.0f;
                if (skip) continue;
                float pj = p[j], mj = m[j], vj = v[j];
                adam1(pj, mj, vj, gj, lr_over_bc1, bc2_sqrt, h);
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (shadow) shadow[j] = (half_t)pj;
            }
        }
    }
}

// multi-tensor forms: one launch walks up to MAX_SEGS tensors (the two MLP weight vectors are ~18 k parameters: a launch
// each would cost more than the work)
constexpr int MAX_SEGS = 8;
struct CheckSegs { const void* grad[MAX_SEGS]; size_t n[MAX_SEGS]; int is_half[MAX_SEGS]; int count; };
struct ApplySegs {
    float* p[MAX_SEGS]; float* m[MAX_SEGS]; float* v[MAX_SEGS]; void* grad[MAX_SEGS]; half_t* shadow[MAX_SEGS];
    const float* lr[MAX_SEGS]; size_t n[MAX_SEGS]; int is_half[MAX_SEGS]; int count;
    const uint32_t* touched[MAX_SEGS];       // bit per 16 parameters (8 table entries x 2): 0 = gradient and both moments are zero
};

template <typename G>
__device__ __forceinline__ bool seg_has_nonfinite(const G* __restrict__ grad, size_t n) {
    bool bad = false;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            if constexpr (sizeof(G) == 2) {
                const uint4 v = *reinterpret_cast<const uint4*>(grad + i);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) bad |= ((w[k] & 0x7c00u) == 0x7c00u) || ((w[k] & 0x7c000000u) == 0x7c000000u);
            } else {
                const float4 a = *reinterpret_cast<const float4*>(grad + i), b = *reinterpret_cast<const float4*>(grad + i + 4);
                bad |= !isfinite(a.x) || !isfinite(a.y) || !isfinite(a.z) || !isfinite(a.w) || !isfinite(b.x) || !isfinite(b.y) ||
                       !isfinite(b.z) || !isfinite(b.w);
            }
        } else {
            for (size_t j = i; j < n; j++) bad |= !isfinite(ldg(grad, j));
        }
    }
    return bad;
}

__global__ __launch_bounds__(256) void k_check_multi(CheckSegs sg, OptState* __restrict__ st) {
    bool bad = false;
    for (int s = 0; s < sg.count; s++)
        bad |= sg.is_half[s] ? seg_has_nonfinite((const half_t*)sg.grad[s], sg.n[s]) : seg_has_nonfinite((const float*)sg.grad[s], sg.n[s]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&st->found_inf, 1);
}

template <typename G>
__device__ __forceinline__ void seg_apply(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, G* __restrict__ grad,
                                          half_t* __restrict__ shadow, size_t n, bool skip, float inv_scale, float bc2_sqrt,
                                          float lr_over_bc1, const AdamHyper& h, const uint32_t* __restrict__ touched = nullptr) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (touched) {
            // a 64-byte line of parameters that never received a gradient has g = exp_avg = exp_avg_sq = 0: Adam's update of it is
            // exactly zero (weight_decay = 0), so nothing of it is read or written (torch.optim.Adam would rewrite the same values)
            const size_t line = i >> 4;
            if (!((touched[line >> 5] >> (line & 31)) & 1u)) continue;
        }
        if (i + 4 <= n) {
            float g[4];
            if constexpr (sizeof(G) == 2) {
                const uint2 raw = *reinterpret_cast<const uint2*>(grad + i);
                const half_t* gh = reinterpret_cast<const half_t*>(&raw);
#pragma unroll
                for (int k = 0; k < 4; k++) g[k] = (float)gh[k];
                *reinterpret_cast<uint2*>(grad + i) = uint2{0u, 0u};
            } else {
                const float4 raw = *reinterpret_cast<const float4*>(grad + i);
                g[0] = raw.x; g[1] = raw.y; g[2] = raw.z; g[3] = raw.w;
                *reinterpret_cast<float4*>(grad + i) = float4{0, 0, 0, 0};
            }
            if (skip) continue;
            float4 pp = *reinterpret_cast<const float4*>(p + i), mm = *reinterpret_cast<const float4*>(m + i),
                   vv = *reinterpret_cast<const float4*>(v + i);
            adam1(pp.x, mm.x, vv.x, g[0] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            adam1(pp.y, mm.y, vv.y, g[1] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            adam1(pp.z, mm.z, vv.z, g[2] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            adam1(pp.w, mm.w, vv.w, g[3] * inv_scale, lr_over_bc1, bc2_sqrt, h);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
            if (shadow) {
                half_t s[4] = {(half_t)pp.x, (half_t)pp.y, (half_t)pp.z, (half_t)pp.w};
                *reinterpret_cast<uint2*>(shadow + i) = *reinterpret_cast<const uint2*>(s);
            }
        } else {
            for (size_t j = i; j < n; j++) {
                const float gj = ldg(grad, j) * inv_scale;
                grad[j] = (G)0.0f;
                if (skip) continue;
                float pj = p[j], mj = m[j], vj = v[j];
                adam1(pj, mj, vj, gj, lr_over_bc1, bc2_sqrt, h);
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (shadow) shadow[j] = (half_t)pj;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_apply_multi(ApplySegs sg, OptState* __restrict__ st, AdamHyper h) {
    const bool skip = st->skip != 0;
    const float inv_scale = st->inv_scale, bc2_sqrt = st->bc2_sqrt;
    // the LAST workgroup's first thread (the first workgroups carry the table's head, the longest work): words 9-13 only
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) prepare_next_step(st, h.beta1, h.beta2);
    for (int s = 0; s < sg.count; s++) {
        const float lr_over_bc1 = (float)((double)sg.lr[s][0] * (double)st->inv_bc1);
        if (sg.is_half[s]) seg_apply(sg.p[s], sg.m[s], sg.v[s], (half_t*)sg.grad[s], sg.shadow[s], sg.n[s], skip, inv_scale, bc2_sqrt, lr_over_bc1, h, sg.touched[s]);
        else seg_apply(sg.p[s], sg.m[s], sg.v[s], (float*)sg.grad[s], sg.shadow[s], sg.n[s], skip, inv_scale, bc2_sqrt, lr_over_bc1, h, sg.touched[s]);
    }
}

// torch_ema.ExponentialMovingAverage.update() (the trainer's `ema`, nerf/utils.py:407-408, 1502-1503):
// shadow -= (1 - decay) * (shadow - param), every parameter tensor in one launch
struct EmaSegs { float* shadow[MAX_SEGS]; const float* param[MAX_SEGS]; size_t n[MAX_SEGS]; int count; };
__global__ __launch_bounds__(256) void k_ema_multi(EmaSegs sg, float one_minus_decay) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (int s = 0; s < sg.count; s++) {
        float* __restrict__ sh = sg.shadow[s];
        const float* __restrict__ p = sg.param[s];
        const size_t n = sg.n[s], n4 = n & ~(size_t)3;
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += stride) {
            float4 a = *reinterpret_cast<const float4*>(sh + i);
            const float4 b = *reinterpret_cast<const float4*>(p + i);
            a.x -= (a.x - b.x) * one_minus_decay; a.y -= (a.y - b.y) * one_minus_decay;
            a.z -= (a.z - b.z) * one_minus_decay; a.w -= (a.w - b.w) * one_minus_decay;
            *reinterpret_cast<float4*>(sh + i) = a;
        }
        if (blockIdx.x == 0 && threadIdx.x < n - n4) { const size_t j = n4 + threadIdx.x; sh[j] -= (sh[j] - p[j]) * one_minus_decay; }
    }
}

uint32_t stream_blocks(size_t n, int per_thread) {
    const size_t want = (n + 256ull * per_thread - 1) / (256ull * per_thread);
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(want, (size_t)lae::num_cus() * 16));
}

}  // namespace

extern "C" {

int lae_adam_check(const void* grad, int grad_is_half, uint64_t n, void* state, void* stream) {
    if (n == 0) return LAE_OK;
    if (!grad || !state) return LAE_ENULL;
    if ((reinterpret_cast<uintptr_t>(grad) & 15) != 0) return LAE_EINVAL;
    OptState* st = reinterpret_cast<OptState*>(state);
    if (grad_is_half) k_check<half_t><<<stream_blocks(n, 8), 256, 0, STREAM(stream)>>>((const half_t*)grad, n, st);
    else k_check<float><<<stream_blocks(n, 8), 256, 0, STREAM(stream)>>>((const float*)grad, n, st);
    return lae::check_launch("adam_check");
}

int lae_adam_check_multi(uint32_t n_tensors, const void* const* grads, const int* grad_is_half, const uint64_t* sizes, void* state,
                         void* stream) {
    if (n_tensors == 0) return LAE_OK;
    if (!grads || !grad_is_half || !sizes || !state) return LAE_ENULL;
    if (n_tensors > (uint32_t)MAX_SEGS) return LAE_EINVAL;
    CheckSegs sg{};
    size_t biggest = 0;
    for (uint32_t i = 0; i < n_tensors; i++) {
        if (sizes[i] && !grads[i]) return LAE_ENULL;
        if (reinterpret_cast<uintptr_t>(grads[i]) & 15) return LAE_EINVAL;
        sg.grad[i] = grads[i]; sg.n[i] = sizes[i]; sg.is_half[i] = grad_is_half[i];
        biggest = std::max(biggest, (size_t)sizes[i]);
    }
    sg.count = (int)n_tensors;
    k_check_multi<<<stream_blocks(biggest, 8), 256, 0, STREAM(stream)>>>(sg, reinterpret_cast<OptState*>(state));
    return lae::check_launch("adam_check_multi");
}

int lae_ema_update_multi(uint32_t n_tensors, float* const* shadows, const float* const* params, const uint64_t* sizes, float one_minus_decay,
                         void* stream) {
    if (n_tensors == 0) return LAE_OK;
    if (!shadows || !params || !sizes) return LAE_ENULL;
    if (n_tensors > (uint32_t)MAX_SEGS) return LAE_EINVAL;
    EmaSegs sg{};
    size_t biggest = 0;
    for (uint32_t i = 0; i < n_tensors; i++) {
        if (sizes[i] && (!shadows[i] || !params[i])) return LAE_ENULL;
        if ((reinterpret_cast<uintptr_t>(shadows[i]) | reinterpret_cast<uintptr_t>(params[i])) & 15) return LAE_EINVAL;
        sg.shadow[i] = shadows[i]; sg.param[i] = params[i]; sg.n[i] = sizes[i];
        biggest = std::max(biggest, (size_t)sizes[i]);
    }
    sg.count = (int)n_tensors;
    k_ema_multi<<<stream_blocks(biggest, 4), 256, 0, STREAM(stream)>>>(sg, one_minus_decay);
    return lae::check_launch("ema_update_multi");
}

int lae_adam_apply_multi(uint32_t n_tensors, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs, void* const* grads,
                         const int* grad_is_half, void* const* shadows_half, const uint64_t* sizes, const float* const* lrs,
                         const void* const* touched_lines, const void* state, float beta1, float beta2, float eps, float weight_decay,
                         void* stream) {
    if (n_tensors == 0) return LAE_OK;
    if (!params || !exp_avgs || !exp_avg_sqs || !grads || !grad_is_half || !shadows_half || !sizes || !lrs || !state) return LAE_ENULL;
    if (n_tensors > (uint32_t)MAX_SEGS) return LAE_EINVAL;
    ApplySegs sg{};
    size_t biggest = 0;
    for (uint32_t i = 0; i < n_tensors; i++) {
        if (sizes[i] && (!params[i] || !exp_avgs[i] || !exp_avg_sqs[i] || !grads[i] || !lrs[i])) return LAE_ENULL;
        const uintptr_t al = reinterpret_cast<uintptr_t>(params[i]) | reinterpret_cast<uintptr_t>(exp_avgs[i]) |
                             reinterpret_cast<uintptr_t>(exp_avg_sqs[i]) | reinterpret_cast<uintptr_t>(grads[i]) |
                             reinterpret_cast<uintptr_t>(shadows_half[i]);
        if (al & 15) return LAE_EINVAL;
        sg.p[i] = params[i]; sg.m[i] = exp_avgs[i]; sg.v[i] = exp_avg_sqs[i]; sg.grad[i] = grads[i];
        sg.shadow[i] = (half_t*)shadows_half[i]; sg.lr[i] = lrs[i]; sg.n[i] = sizes[i]; sg.is_half[i] = grad_is_half[i];
        sg.touched[i] = (touched_lines && weight_decay == 0.0f) ? (const uint32_t*)touched_lines[i] : nullptr;
        biggest = std::max(biggest, (size_t)sizes[i]);
    }
    sg.count = (int)n_tensors;
    k_apply_multi<<<stream_blocks(biggest, 4), 256, 0, STREAM(stream)>>>(sg, reinterpret_cast<OptState*>(const_cast<void*>(state)),
                                                                        AdamHyper{beta1, beta2, eps, weight_decay});
    return lae::check_launch("adam_apply_multi");
}

int lae_adam_begin(void* state, float beta1, float beta2, int growth_interval, float growth_factor, float backoff_factor,
                   int use_scaler, void* stream) {
    if (!state) return LAE_ENULL;
    k_begin<<<1, 64, 0, STREAM(stream)>>>(reinterpret_cast<OptState*>(state), beta1, beta2, growth_interval, growth_factor,
                                          backoff_factor, use_scaler);
    return lae::check_launch("adam_begin");
}

int lae_adam_apply(float* param, float* exp_avg, float* exp_avg_sq, void* grad, int grad_is_half, void* shadow_half, uint64_t n,
                   const void* state, const float* lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (n == 0) return LAE_OK;
    if (!param || !exp_avg || !exp_avg_sq || !grad || !state || !lr) return LAE_ENULL;
    const uintptr_t al = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq) |
                         reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(shadow_half);
    if (al & 15) return LAE_EINVAL;
    const AdamHyper h{beta1, beta2, eps, weight_decay};
    OptState* st = reinterpret_cast<OptState*>(const_cast<void*>(state));
    if (grad_is_half)
        k_apply<half_t><<<stream_blocks(n, 4), 256, 0, STREAM(stream)>>>(param, exp_avg, exp_avg_sq, (half_t*)grad, (half_t*)shadow_half, n, st, lr, h);
    else
        k_apply<float><<<stream_blocks(n, 4), 256, 0, STREAM(stream)>>>(param, exp_avg, exp_avg_sq, (float*)grad, (half_t*)shadow_half, n, st, lr, h);
    return lae::check_launch("adam_apply");
}

}  // extern "C"
