// editgrid.hip -- region growing of LAENeRF's edit grid on the device.
//
// Reference: EditGrid.grow_region_queue (editing/editgrid.py:274-340), a Python loop over a collections.deque: pop up
// to 32 cells, look their density and selection bit up, select the accepted ones, push their 6 face neighbours, repeat
// until `grow_iterations` cells were popped -- ~20 tensor ops, a .cuda() allocation pair and 32 popleft() per round,
// ~160 rounds per call from the GUI.
//
// The selection depends on the exact FIFO order (the pop budget cuts the flood fill short, duplicates count as pops,
// the level of pushed cells is the level of the batch's first cell), so this is a faithful emulation, not a
// wavefront BFS: ONE wavefront walks the queue batch by batch; inside a batch the 32 cells are tested in parallel, the
// survivors' neighbours are ranked with a wave scan and appended in the reference's order.  Bits are written with the
// semantics of the reference's indexed byte assignment `bitfield[i] = (bitfield[i] & ~mask) | bit` evaluated on CPU
// tensors: all right-hand sides are read first, and when several accepted cells of a batch share a byte the LAST one
// wins (the others' bits are not set by this batch).  On CUDA that assignment is a write race; the CPU order is the
// deterministic member of its outcomes and what tests/golden/editgrid.npz was captured with.
#include "lae_common.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

__device__ __forceinline__ uint32_t spread10(uint32_t v) {   // raymarching.cu:56-63
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ uint8_t ld_byte(const uint8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_byte(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// queue entry: x | y << 8 | z << 16 | level << 24 (coordinates < 128 by construction: neighbours are bounds-filtered)
// state: [0] head, [1] tail, [2] cells popped by this call, [3] overflow flag
__global__ __launch_bounds__(64) void k_grow_region(uint8_t* __restrict__ grid, const float* __restrict__ density, uint32_t H,
                                                    float thresh, uint32_t* __restrict__ queue, uint32_t cap,
                                                    uint32_t* __restrict__ state, uint32_t grow_iterations, uint32_t max_n) {
    const int lane = threadIdx.x;
    const uint32_t V = H * H * H;
    uint32_t head = state[0], tail = state[1], ctr = 0, overflow = 0;
    const int dx[6] = {-1, 0, 0, 0, 0, 1}, dy[6] = {0, -1, 0, 0, 1, 0}, dz[6] = {0, 0, -1, 1, 0, 0};      // :314-321
    while (ctr < grow_iterations && head != tail) {
        const uint32_t num = min(min(max_n, tail - head), grow_iterations - ctr);                          // :289
        const bool have = (uint32_t)lane < num;
        const uint32_t e = have ? ld_u32(queue + head + lane) : 0u;
        const int x = (int)(e & 0xffu), y = (int)((e >> 8) & 0xffu), z = (int)((e >> 16) & 0xffu);
        const uint32_t lvl = e >> 24;
        const uint32_t pos = (spread10((uint32_t)x) | (spread10((uint32_t)y) << 1) | (spread10((uint32_t)z) << 2)) % V;   // :299-301
        const uint32_t byte_i = pos / 8 + (V * lvl) / 8, bit = pos % 8;
        const float dens = have ? density[(size_t)lvl * V + pos] : 0.0f;
        const uint8_t old = have ? ld_byte(grid + byte_i) : (uint8_t)0;
        const bool cond = have && dens >= thresh && !((old >> bit) & 1u);                                  // :303-308
        const unsigned long long cmask = __ballot(cond);
        const uint32_t lvl0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lvl);                          // :323 `lvl[0]`
        if (cmask) {
            // indexed byte assignment, last accepted cell of a byte wins (see the header)
            bool is_last = cond;
            for (uint32_t j = 1; j < num; j++) {
                const uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)byte_i, (int)j);
                if (((cmask >> j) & 1ull) && (int)j > lane && bj == byte_i) is_last = false;
            }
            if (is_last) st_byte(grid + byte_i, (uint8_t)((old & ~(1u << bit)) | (1u << bit)));            // editgrid.py:35-38
            // neighbours of the accepted cells, in cell order then offset order, bounds-filtered (:314-331)
            uint32_t vmask = 0;
            if (cond) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int nx = x + dx[k], ny = y + dy[k], nz = z + dz[k];
                    if (nx >= 0 && ny >= 0 && nz >= 0 && nx < (int)H && ny < (int)H && nz < (int)H) vmask |= 1u << k;
                }
            }
            const uint32_t cnt = (uint32_t)__popc(vmask);
            const uint32_t incl = lae::wave_incl_scan(cnt);
            const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
            if (tail + total > cap) { overflow = 1; break; }
            uint32_t at = tail + incl - cnt;
#pragma unroll
            for (int k = 0; k < 6; k++)
                if ((vmask >> k) & 1u)
                    st_u32(queue + at++, (uint32_t)(x + dx[k]) | ((uint32_t)(y + dy[k]) << 8) | ((uint32_t)(z + dz[k]) << 16) | (lvl0 << 24));
            tail += total;
        }
        __threadfence();
        head += num;
        ctr += num;
    }
    if (lane == 0) { state[0] = head; state[1] = tail; state[2] = ctr; state[3] = overflow; }
}

}  // namespace

extern "C" {

int lae_grow_region(uint8_t* grid, const float* density_grid, uint32_t C, uint32_t H, float density_thresh, uint32_t* queue,
                    uint32_t capacity, uint32_t* state, uint32_t grow_iterations, uint32_t max_batch, void* stream) {
    if (!grid || !density_grid || !queue || !state) return LAE_ENULL;
    if (C == 0 || C > 8 || H == 0 || H > 128 || max_batch == 0 || max_batch > 32) return LAE_EINVAL;
    if (grow_iterations == 0) return LAE_OK;
    k_grow_region<<<1, 64, 0, STREAM(stream)>>>(grid, density_grid, H, density_thresh, queue, capacity, state, grow_iterations, max_batch);
    return lae::check_launch("grow_region");
}

}  // extern "C"
