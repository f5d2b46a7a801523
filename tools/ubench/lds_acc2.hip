// microbenchmark (round 2): cost of the accumulate pass's LDS updates on gfx950.
// 1024-thread workgroup, 8192-entry partition, items = (e, v) streamed from global memory (coalesced).
//  mode 0: 2 x ds_add_u64, AoS accumulators acc[2e], acc[2e+1]                      (round-1 kernel)
//  mode 1: 2 x ds_add_u64, SoA accumulators a0[e], a1[e]
//  mode 2: 1 x ds_add_u64 (half the work; isolates per-instruction cost)
//  mode 3: 2 x ds_add_u32 SoA
//  mode 4: plain ds_read_b128 + add + ds_write_b128 (NOT atomic: upper bound for a claim-based scheme)
//  mode 5: ds_add_rtn_u64 x2 (returning)
//  mode 6: no LDS op (stream only)
// patterns: 0 random e; 1 e sorted so that consecutive lanes have consecutive (e mod 32) (bank-conflict free for SoA u64);
//           2 runs of 4 equal e; 3 all the same e
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
constexpr uint32_t REPS = 16;   // the same 64 KiB of items again and again (L1 / L2 hits): LDS-side cost, not the stream
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint2* __restrict__ items, uint32_t n, uint32_t* out) {
    __shared__ unsigned long long acc[16384];
    for (uint32_t i = threadIdx.x; i < 16384; i += 1024) acc[i] = 0;
    __syncthreads();
    const uint2* my = items + (size_t)blockIdx.x * n;
    uint32_t dummy = 0;
    for (uint32_t rep = 0; rep < REPS; rep++)
    for (uint32_t i = threadIdx.x; i + 3 * 1024 < n; i += 4 * 1024) {
        uint2 it[4] = {my[i], my[i + 1024], my[i + 2048], my[i + 3072]};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t e = it[q].x & 8191u;
            const unsigned long long v0 = it[q].y & 0xffffu, v1 = it[q].y >> 16;
            if (MODE == 0) { atomicAdd(&acc[2 * e], v0); atomicAdd(&acc[2 * e + 1], v1); }
            else if (MODE == 1) { atomicAdd(&acc[e], v0); atomicAdd(&acc[8192 + e], v1); }
            else if (MODE == 2) { atomicAdd(&acc[e], v0); }
            else if (MODE == 3) { uint32_t* a = reinterpret_cast<uint32_t*>(acc); atomicAdd(&a[e], (uint32_t)v0); atomicAdd(&a[8192 + e], (uint32_t)v1); }
            else if (MODE == 4) { ulonglong2* a = reinterpret_cast<ulonglong2*>(acc); ulonglong2 t = a[e]; t.x += v0; t.y += v1; a[e] = t; }
            else if (MODE == 5) { dummy += (uint32_t)atomicAdd(&acc[e], v0); dummy += (uint32_t)atomicAdd(&acc[8192 + e], v1); }
            else dummy += e + (uint32_t)v0;
        }
    }
    __syncthreads();
    uint32_t s = dummy;
    for (uint32_t i = threadIdx.x; i < 16384; i += 1024) s += (uint32_t)acc[i];
    if (s == 0xdeadbeef) out[0] = s;
}
int main() {
    const uint32_t n = 8192, nb = 256 * 3;
    std::vector<uint2> h((size_t)n * nb);
    uint2* d; uint32_t* o; hipMalloc(&d, h.size() * 8); hipMalloc(&o, 64);
    const char* pn[4] = {"random", "bank-sorted", "runs of 4", "all same"};
    const char* mn[7] = {"2x ds_add_u64 AoS", "2x ds_add_u64 SoA", "1x ds_add_u64", "2x ds_add_u32 SoA", "b128 read+write (non-atomic)", "2x ds_add_rtn_u64", "stream only"};
    for (int pattern = 0; pattern < 4; pattern++) {
        uint32_t x = 12345;
        for (size_t i = 0; i < h.size(); i++) {
            x = x * 1664525u + 1013904223u;
            uint32_t e = pattern == 0 ? (x >> 8) & 8191u : pattern == 1 ? (((x >> 8) & 8191u) & ~31u) | (uint32_t)(i & 31) : pattern == 2 ? ((uint32_t)(i / 4) * 2654435761u >> 8) & 8191u : 7u;
            h[i] = make_uint2(e, x);
        }
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 7; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: k<0><<<nb, 1024>>>(d, n, o); break; case 1: k<1><<<nb, 1024>>>(d, n, o); break; case 2: k<2><<<nb, 1024>>>(d, n, o); break;
                    case 3: k<3><<<nb, 1024>>>(d, n, o); break; case 4: k<4><<<nb, 1024>>>(d, n, o); break; case 5: k<5><<<nb, 1024>>>(d, n, o); break;
                    default: k<6><<<nb, 1024>>>(d, n, o); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            // 3 blocks per CU in sequence (1 resident: 128 KiB LDS) -> per-CU items = 3 * n
            printf("pattern %-11s mode %-30s: %7.1f us -> %.3f ns per item per CU (%.2f cycles @2.4GHz)\n", pn[pattern], mn[mode], best * 1e3,
                   best * 1e6 / (3.0 * n * REPS), best * 1e6 / (3.0 * n * REPS) * 2.4);
        }
    }
    return 0;
}
