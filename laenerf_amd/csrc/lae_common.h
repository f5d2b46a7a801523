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
                       const int32_t* offsets_host, uint32_t B_likely = 0);   // B_likely: rows the host expects (0: B_launch); sizes launches, never a result
// Loop state of the frame loop, double-buffered in device memory (frame.hip k_frame_emit advances it).
// Sample rows are RAY-MAJOR IN 64-ROW GROUPS: a group holds rpg = 64 / n_step whole rays (integer division), ray n's row j is
// (n / rpg) * 64 + (n % rpg) * n_step + j; the 64 - rpg * n_step rows at a group's end (n_step = 3, 5, 6, 7 only) are padding.
// A wave of the head kernel owns a group, so the compositing of a ray never crosses waves.
struct FrameCtrl { uint32_t n_alive, n_step, n_rows, step, iter, done, total_rows, pad; };
static_assert(sizeof(FrameCtrl) == 32, "FrameCtrl layout");
__host__ __device__ __forceinline__ uint32_t frame_rays_per_group(uint32_t n_step) { return 64u / n_step; }
// per-ray accumulators of composite_rays (raymarching.cu:948-1035 / :1037-1142) as ONE 32-byte record: two 16-byte
// loads / stores per ray and iteration instead of six to eight scattered 4-byte ones
struct __attribute__((aligned(16))) RayAcc { float ws, depth, r, g, b, t, wse, de; };
static_assert(sizeof(RayAcc) == 32, "RayAcc layout");
struct FrameHeadArgs {
    const FrameCtrl* cur;            // this iteration's state
    const int32_t* alive;            // compact list of ray indices
    const float* deltas;             // [rows, 2]
    const uint8_t* edit_occ;         // [rows] (distill render) or NULL
    RayAcc* acc;                     // [N]
    int32_t* seg_next;               // survivors of wave u (unit), in order, at [u * R, u * R + seg_counts_next[u])
    uint32_t* seg_counts_next;       // per wave (unit)
    uint32_t* blk_counts_next;       // per workgroup: the sum of its FRAME_HEAD_WAVES unit counts
    uint32_t R;
    float T_thresh;
};
// head + compositing of one iteration in one launch of n_blocks workgroups x FRAME_HEAD_WAVES waves; every wave is a
// unit of work (a contiguous run of 64-row groups) and writes one survivor segment (stride fa.R) + its count
constexpr int FRAME_HEAD_WAVES = 8;
int nerf_head_composite_frame(const void* enc, const float* dirs, const void* sigma_weights, const void* color_weights,
                              uint32_t M_cap, float density_scale, const FrameHeadArgs& fa, bool edit, uint32_t n_blocks,
                              hipStream_t stream);
uint32_t frame_head_max_blocks();

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
