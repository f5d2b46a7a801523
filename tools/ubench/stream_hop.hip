// stream_hop.hip -- what does one cross-stream dependency cost on this box?  (round 4, frame loop: the per-iteration
// handshake between the caller's stream and the lookahead stream.)  Two streams play ping-pong with a work kernel of a fixed
// duration; the hop cost is (total - hops * work) / hops.  Mechanisms:
//   event      hipEventRecord + hipStreamWaitEvent (flags: timing disabled / + DisableSystemFence / + ReleaseToDevice)
//   kernel     a one-thread kernel sets a flag in device memory, a one-wave kernel on the other stream polls it
//   inkernel   the work kernel's last-dispatched block sets the flag itself (no extra dispatch on the producer side)
//   memop      hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory
//   same       both kernels in ONE stream (the floor: no hop at all)
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/stream_hop.hip -o tools/ubench/bin/stream_hop ; run: stream_hop [work_us] [hops]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_work(unsigned long long ticks, unsigned* flag, unsigned value) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (flag && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_set(unsigned* flag, unsigned value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(const unsigned* flag, unsigned value) {
    for (unsigned spin = 0; spin < (1u << 24); spin++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= value) return;
        __builtin_amdgcn_s_sleep(2);
    }
}

int main(int argc, char** argv) {
    const double work_us = argc > 1 ? atof(argv[1]) : 20.0;
    const int hops = argc > 2 ? atoi(argv[2]) : 400;
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    const unsigned long long ticks = (unsigned long long)(work_us * clk_khz / 1000.0);
    int can_wait = 0;
    CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    unsigned* flag = nullptr;
    CK(hipMalloc(&flag, 256));
    unsigned* sig = nullptr;
    const bool have_sig = can_wait && hipExtMallocWithFlags((void**)&sig, 256, hipMallocSignalMemory) == hipSuccess;
    const int blocks = 256;
    auto run = [&](const char* name, auto&& hop) {
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(flag, 0, 256));
            if (have_sig) CK(hipMemset(sig, 0, 256));
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < hops; i++) hop(i, s[i & 1], s[(i + 1) & 1]);
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best;
        }
        printf("%-28s %8.1f us total, %6.2f us per hop beyond the %.0f us of work\n", name, best, best / hops - work_us, work_us);
    };
    std::vector<hipEvent_t> ev(64);
    for (unsigned fl : {hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence, hipEventDisableTiming | hipEventReleaseToDevice}) {
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, fl));
        char name[64];
        snprintf(name, sizeof name, "event (flags 0x%x)", fl);
        run(name, [&](int i, hipStream_t a, hipStream_t b) {
            hipLaunchKernelGGL(k_work, dim3(blocks), dim3(64), 0, a, ticks, (unsigned*)nullptr, 0u);
            CK(hipEventRecord(ev[i & 63], a));
            CK(hipStreamWaitEvent(b, ev[i & 63], 0));
        });
        for (auto& e : ev) CK(hipEventDestroy(e));
    }
    run("kernel (set + poll)", [&](int i, hipStream_t a, hipStream_t b) {
        hipLaunchKernelGGL(k_work, dim3(blocks), dim3(64), 0, a, ticks, (unsigned*)nullptr, 0u);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, a, flag, (unsigned)(i + 1));
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, b, flag, (unsigned)(i + 1));
    });
    run("inkernel set + poll kernel", [&](int i, hipStream_t a, hipStream_t b) {
        hipLaunchKernelGGL(k_work, dim3(blocks), dim3(64), 0, a, ticks, flag, (unsigned)(i + 1));
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, b, flag, (unsigned)(i + 1));
    });
    if (have_sig) {
        run("memop (write + wait value)", [&](int i, hipStream_t a, hipStream_t b) {
            hipLaunchKernelGGL(k_work, dim3(blocks), dim3(64), 0, a, ticks, (unsigned*)nullptr, 0u);
            CK(hipStreamWriteValue32(a, sig, (unsigned)(i + 1), 0));
            CK(hipStreamWaitValue32(b, sig, (unsigned)(i + 1), hipStreamWaitValueGte, 0xffffffffu));
        });
    } else printf("memop: not available (CanUseStreamWaitValue = %d)\n", can_wait);
    run("same stream (no hop)", [&](int, hipStream_t, hipStream_t) {
        hipLaunchKernelGGL(k_work, dim3(blocks), dim3(64), 0, s[0], ticks, (unsigned*)nullptr, 0u);
    });
    run("same stream + tiny kernel", [&](int i, hipStream_t, hipStream_t) {
        hipLaunchKernelGGL(k_work, dim3(blocks), dim3(64), 0, s[0], ticks, (unsigned*)nullptr, 0u);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s[0], flag, (unsigned)(i + 1));
    });
    return 0;
}
