// Does a packed-fp32 product read back by the NEXT packed-fp32 instruction need a wait state on gfx950?
// LLVM puts `s_nop 0` between such a pair only when op_sel_hi[0] of the producer is set (its hazard recogniser reads that bit
// as VOP3's DST_OP_SEL), so `v_pk_mul_f32 vT, vA, vB op_sel:[0,1] op_sel_hi:[0,1]` + `v_pk_mul_f32 vR, vT, vC` is emitted back to
// back -- the one place where round 4's failing "loop form" of k_grid_fwd_lean differs from the passing one.
// Every lane runs the pair without and with the nop on the same inputs and counts results that differ.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/pk_hazard tools/ubench/pk_hazard.hip ; pk_hazard [seconds] [mode] (beside a neighbour: a second copy, tools/ubench/bin/spinner mfma, ...)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

// mode 0: the pair back to back against the pair with s_nop (round 5: 7.4e13 pairs, 0 differences -- not the cause)
// mode 1: `v_pk_mul_f32 vT, vA, vB op_sel:[0,1] op_sel_hi:[0,1]` (both result halves = A.lo * B.hi, A.lo broadcast through SRC0:
//         the instruction of the failing kernel) against the same product with the operands swapped,
//         `v_pk_mul_f32 vT, vB, vA op_sel:[1,0] op_sel_hi:[1,0]` (broadcast through SRC1: the passing kernel's)
__global__ __launch_bounds__(256) void k_pair(const float* __restrict__ in, unsigned long long* __restrict__ errs, int iters, float* __restrict__ sink, int mode) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    f2 a = {in[tid * 6 + 0], in[tid * 6 + 1]}, b = {in[tid * 6 + 2], in[tid * 6 + 3]}, c = {in[tid * 6 + 4], in[tid * 6 + 5]};
    unsigned bad = 0;
    f2 acc = {0.f, 0.f};
    for (int i = 0; i < iters; i++) {
        f2 t0, r0, t1, r1;
        if (mode == 0) {
            asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                         "v_pk_mul_f32 %1, %0, %4"
                         : "=&v"(t0), "=&v"(r0) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                         "s_nop 0\n\t"
                         "v_pk_mul_f32 %1, %0, %4"
                         : "=&v"(t1), "=&v"(r1) : "v"(a), "v"(b), "v"(c));
        } else {
            asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                         "v_pk_mul_f32 %1, %0, %4"
                         : "=&v"(t0), "=&v"(r0) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_pk_mul_f32 %0, %3, %2 op_sel:[1,0] op_sel_hi:[1,0]\n\t"
                         "s_nop 0\n\t"
                         "v_pk_mul_f32 %1, %0, %4"
                         : "=&v"(t1), "=&v"(r1) : "v"(a), "v"(b), "v"(c));
        }
        bad += (__builtin_bit_cast(unsigned long long, r0) != __builtin_bit_cast(unsigned long long, r1)) ? 1u : 0u;
        acc += r0;
        a.x = a.x * 1.0009765625f + 0.001f; a.y = a.y * 0.9990234375f - 0.002f;
        b.y = b.y * 1.001953125f + 0.003f; c.x = c.x * 0.998046875f + 0.004f; c.y = c.y * 1.00048828125f - 0.005f;
    }
    if (bad) atomicAdd(errs, (unsigned long long)bad);
    if (acc.x == 123.456f) sink[tid] = acc.y;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int mode = argc > 2 ? atoi(argv[2]) : 1;
    const unsigned n = 4096 * 256;
    float* h = (float*)malloc(n * 6 * sizeof(float));
    srand(7);
    for (unsigned i = 0; i < n * 6; i++) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *d, *sink; unsigned long long* errs;
    if (hipMalloc(&d, n * 6 * sizeof(float)) != hipSuccess || hipMalloc(&sink, n * sizeof(float)) != hipSuccess || hipMalloc(&errs, 8) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, n * 6 * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemset(errs, 0, 8);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long launches = 0;
    bool said = false;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 16; k++) { k_pair<<<4096, 256>>>(d, errs, 256, sink, mode); launches++; }
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        if (!said) { printf("ready\n"); fflush(stdout); said = true; }
    }
    unsigned long long e = 0;
    (void)hipMemcpy(&e, errs, 8, hipMemcpyDeviceToHost);
    printf("{\"mode\": %d, \"pairs\": %llu, \"pairs_that_differ\": %llu}\n", mode, launches * n * 256ull, e);
    return 0;
}
