// Neighbour process for tools/grid_loop_fault.py: keeps the GPU busy from a SECOND process until it is killed or its time is up.
//   spinner alu <seconds>     : 2048 workgroups of pure VALU work per launch (no memory traffic beyond one store per wave)
//   spinner stream <seconds>  : copies a 1 GiB buffer back and forth (HBM / L2 traffic, little ALU)
//   spinner mfma <seconds>    : v_mfma_f32_16x16x32_f16 back to back on every SIMD (matrix pipe busy, no memory traffic)
//   spinner mfma32 <seconds>  : v_mfma_f32_32x32x16_f16 back to back;  spinner mfmaf32: v_mfma_f32_16x16x4_f32;  spinner mfmabf16: v_mfma_f32_16x16x32_bf16
//   spinner lds <seconds>     : ds_read / ds_write loops (LDS pipe busy)
// Prints "ready" after its first launch completed.  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/spinner tools/ubench/spinner.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void k_alu(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    for (int i = 0; i < iters; i++) { a = fmaf(a, 1.0001f, b); b = fmaf(b, 0.9999f, a); }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = a + b;
}
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(blockIdx.x * 1e-4f - i); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = c0[0] + c1[1] + c2[2] + c3[3];
}
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void k_mfma_other(float* out, int iters) {
    h8 a, b; b8 ab, bb;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(blockIdx.x * 1e-4f - i); ab[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; }
    f16v c32 = {}; f4 c0 = {0, 0, 0, 0}, c1 = c0;
    for (int i = 0; i < iters; i++) {
        if constexpr (KIND == 0) { c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c32, 0, 0, 0); c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c32, 0, 0, 0); }
        if constexpr (KIND == 1) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[1], (float)b[1], c1, 0, 0, 0); }
        if constexpr (KIND == 2) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ab, c1, 0, 0, 0); }
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = c32[0] + c0[0] + c1[1];
}
__global__ void k_lds(float* out, int iters) {
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = i;
    __syncthreads();
    float acc = 0.f;
    unsigned j = threadIdx.x;
    for (int i = 0; i < iters; i++) { acc += sh[j & 4095u]; sh[(j * 7u + 3u) & 4095u] = acc; j = j * 5u + 1u; }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = acc;
}
__global__ void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: spinner alu|stream|mfma|lds seconds\n"); return 2; }
    const bool mfma = strcmp(argv[1], "mfma") == 0, lds = strcmp(argv[1], "lds") == 0;
    const int other = strcmp(argv[1], "mfma32") == 0 ? 0 : strcmp(argv[1], "mfmaf32") == 0 ? 1 : strcmp(argv[1], "mfmabf16") == 0 ? 2 : -1;
    const bool alu = strcmp(argv[1], "alu") == 0 || mfma || lds || other >= 0;
    const double seconds = atof(argv[2]);
    float* out = nullptr; float4 *a = nullptr, *b = nullptr;
    const size_t n = (1ull << 30) / sizeof(float4);
    if (hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    if (!alu && (hipMalloc(&a, n * sizeof(float4)) != hipSuccess || hipMalloc(&b, n * sizeof(float4)) != hipSuccess)) return 1;
    if (!alu) (void)hipMemset(a, 1, n * sizeof(float4));
    const auto t0 = std::chrono::steady_clock::now();
    bool said = false;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 8; k++) {
            if (other == 0) k_mfma_other<0><<<2048, 256>>>(out, 10000);
            else if (other == 1) k_mfma_other<1><<<2048, 256>>>(out, 20000);
            else if (other == 2) k_mfma_other<2><<<2048, 256>>>(out, 20000);
            else if (mfma) k_mfma<<<2048, 256>>>(out, 20000);
            else if (lds) k_lds<<<2048, 256>>>(out, 20000);
            else if (alu) k_alu<<<2048, 256>>>(out, 20000);
            else { k_copy<<<4096, 256>>>(a, b, n); k_copy<<<4096, 256>>>(b, a, n); }
        }
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        if (!said) { printf("ready\n"); fflush(stdout); said = true; }
    }
    return 0;
}
