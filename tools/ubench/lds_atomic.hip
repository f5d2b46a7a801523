// microbenchmark: LDS float-atomic throughput on gfx950 (random vs same-address, pk_f16 vs f32 vs plain store)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int MODE>  // 0 pk_add_f16, 1 add_f32, 2 plain ds_write, 3 pk_add_f16 via __shared__ direct
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ idx, uint32_t n, uint32_t* out) {
    __shared__ uint32_t acc[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) acc[i] = 0;
    __syncthreads();
    const uint32_t* my = idx + (size_t)blockIdx.x * n;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t e = my[i] & 32767u;
        if (MODE == 0) { h2 v = {(_Float16)1.0f, (_Float16)0.5f}; __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)(acc) + e, v); }
        else if (MODE == 1) atomicAdd(reinterpret_cast<float*>(acc) + e, 1.0f);
        else if (MODE == 2) acc[e] = i;
        else if (MODE == 3) atomicAdd(acc + e, 3u);
        else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (e & 16383u), 3ull);
        else if (MODE == 5) { atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (e & 16382u), 3ull); atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (e & 16382u) + 1, 5ull); }
    }
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) s += acc[i];
    if (s == 0xdeadbeef) out[0] = s;
}
int main() {
    const uint32_t n = 131072, nb = 176;
    uint32_t* h = (uint32_t*)malloc((size_t)n * nb * 4);
    uint32_t *d, *o; hipMalloc(&d, (size_t)n * nb * 4); hipMalloc(&o, 64);
    for (int pattern = 0; pattern < 3; pattern++) {
        uint32_t x = 12345;
        for (size_t i = 0; i < (size_t)n * nb; i++) {
            x = x * 1664525u + 1013904223u;
            h[i] = pattern == 0 ? (x >> 8) : pattern == 1 ? (uint32_t)(i / 8) * 2654435761u >> 8 : 7u;   // random | runs of 8 equal | all same
        }
        hipMemcpy(d, h, (size_t)n * nb * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 6; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<nb, 1024>>>(d, n, o); else if (mode == 1) k<1><<<nb, 1024>>>(d, n, o); else if (mode == 2) k<2><<<nb, 1024>>>(d, n, o); else if (mode == 3) k<3><<<nb, 1024>>>(d, n, o); else if (mode == 4) k<4><<<nb, 1024>>>(d, n, o); else k<5><<<nb, 1024>>>(d, n, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("pattern %d (%s) mode %d (%s): %.1f us for %u ops/block -> %.2f cycles/lane-op @2.4GHz\n", pattern,
                   pattern == 0 ? "random" : pattern == 1 ? "runs of 8" : "all same", mode, mode == 0 ? "ds_pk_add_f16" : mode == 1 ? "ds_add_f32" : mode == 2 ? "ds_write" : mode == 3 ? "ds_add_u32" : mode == 4 ? "ds_add_u64" : "2x ds_add_u64 adjacent",
                   ms * 1e3, n, ms * 1e-3 * 2.4e9 / n);
        }
    }
    return 0;
}
