// lae_common.h -- shared host/device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/laenerf.h"

#define LAE_WAVE 64

namespace lae {

void set_last_error(const char* what, hipError_t e);

// Library-owned, grow-only device workspaces (the counterpart of the reference's process-global split-K
// streams + CUTLASS workspace, ffmlp.cu:711-740, cutlass_matmul.h:335-352), one set per device.  Returns nullptr on
// failure (error string set).  Steady state is allocation free; growth allocates a new buffer WITHOUT freeing the old
// one (queued kernels and captured graphs may still use it) and is refused while `stream` is being captured.
enum WsSlot { WS_FFMLP_SLABS = 0, WS_GRID_GRAD_T = 1, WS_GRID_OUT_T = 2, WS_GRID_BINS = 3, WS_SLOTS = 4 };
void* workspace(WsSlot slot, size_t bytes, hipStream_t stream);
size_t workspace_bytes(bool retired);
uint64_t workspace_epoch();      // changes whenever free_workspaces() ran (a workspace address may then be a new allocation)
void free_workspaces();
int num_cus();
void set_last_error_str(const char* what);

// entry points of the whole-frame inference loop into the encoder / MLP translation units (lae_render_frame,
// raymarching.hip): same kernels as the C ABI, live row count read from device memory
int grid_forward_frame(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B_cap,
                       uint32_t B_launch, const uint32_t* B_dev, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                       int align_corners, uint32_t interp, float in_shift, float in_scale, hipStream_t stream,
                       const int32_t* offsets_host);
int nerf_head_forward_frame(const void* enc, const float* dirs, const void* sigma_weights, const void* color_weights,
                            uint32_t M_cap, uint32_t M_launch, const uint32_t* n_rows_dev, float density_scale, float* sigmas,
                            float* rgbs, hipStream_t stream);

// every launch is followed by this: the reference never checked its launches
// (SURVEY 8b "Errors"); we do, and surface the error through the return code.
static inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error(what, e); return LAE_ELAUNCH; }
    return LAE_OK;
}

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// wave64 inclusive scan on the DPP data path: row shifts 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 / :31
// across them (GFX9 controls); lanes a shift leaves without a source add 0.  (__shfl_up compiles to ds_bpermute_b32, an
// LDS round trip per step.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp_u32<0x111, 0xf>(v); v += dpp_u32<0x112, 0xf>(v); v += dpp_u32<0x114, 0xf>(v); v += dpp_u32<0x118, 0xf>(v);
    v += dpp_u32<0x142, 0xa>(v);
    v += dpp_u32<0x143, 0xc>(v);
    return v;
}

// block-wide exclusive scan for blocks of NW waves; returns exclusive prefix, *total = block sum.
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* lds /*NW+1*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) { uint32_t t = lds[i]; lds[i] = s; s += t; }
        lds[NW] = s;
    }
    __syncthreads();
    uint32_t r = inc - v + lds[w];
    *total = lds[NW];
    __syncthreads();
    return r;
}

}  // namespace lae
