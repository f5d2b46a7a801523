// microbenchmark (round 4): are global float atomics cheap when ONE XCD owns the addresses?
// The hash-grid backward of the reference is 8 x 16 atomicAdd(half2) per sample (gridencoder.cu:248-340).  Round 1 measured
// scattered device-scope float atomics at ~20 G/s chip-wide on gfx950 (they execute at the memory side: the XCDs' L2s are not
// coherent with each other), which is why the shipped backward sorts the contributions instead.  The forward pins a level's
// table to one XCD (blockIdx & 7).  If the backward did the same, no other XCD would touch a level's gradient table during the
// kernel and the atomics would not need device scope: without sc1 they can execute in the XCD's own L2.  How fast is that?
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_atomic.hip -o tools/ubench/bin/l2_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

// MODE 0: global_atomic_pk_add_f16, no scope bits (wavefront scope: executes in the local L2)
// MODE 1: the same with sc1 (device scope)
// MODE 2: global_atomic_add_f32, no scope bits
// MODE 3: global_atomic_add_f32 sc1
// MODE 4: plain 4-byte load of the same addresses (the forward's access pattern, for scale)
// PIN: blocks of XCD x (blockIdx & 7) work on table x only; otherwise block b works on table (b >> 3) & 7 (any XCD, any table)
template <int MODE, bool PIN>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ idx, uint32_t n_per_table, uint32_t* __restrict__ tables,
                                         uint32_t table_words, uint32_t* __restrict__ sink) {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t t = PIN ? xcd : (j & 7u);
    const uint32_t slot = PIN ? j : ((j >> 3) * 8u + xcd);            // both enumerate n_per_table / 256 blocks per table
    const uint32_t i = slot * 256u + threadIdx.x;
    if (i >= n_per_table) return;
    const uint32_t e = idx[(size_t)t * n_per_table + i] % table_words;
    uint32_t* p = tables + (size_t)t * table_words + e;
    if (MODE == 0) { uint32_t v = 0x38003c00u; asm volatile("global_atomic_pk_add_f16 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
    if (MODE == 1) { uint32_t v = 0x38003c00u; asm volatile("global_atomic_pk_add_f16 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
    if (MODE == 2) { float v = 1.0f; asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
    if (MODE == 3) { float v = 1.0f; asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
    if (MODE == 4) { const uint32_t v = *p; if (v == 0xdeadbeefu) sink[0] = v; }
}

int main() {
    const uint32_t table_words = 1u << 19;                 // one hashed level: 2^19 half2 entries = 2 MB (or 2^19 floats)
    const uint32_t n_per_table = 257792u * 8u;             // the bench batch: 8 corners per sample and level
    std::vector<uint32_t> h((size_t)8 * n_per_table);
    uint32_t *d_idx, *d_tab, *d_sink;
    hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_tab, (size_t)8 * table_words * 4); hipMalloc(&d_sink, 64);
    for (int pattern = 0; pattern < 3; pattern++) {
        uint32_t x = 12345;
        for (size_t i = 0; i < h.size(); i++) {
            x = x * 1664525u + 1013904223u;
            // random | x-neighbour pairs (idx, idx ^ 1: the two corners of a row share a line) | runs of 8 samples in one cell
            h[i] = pattern == 0 ? (x >> 8) : pattern == 1 ? ((uint32_t)(i / 2) * 2654435761u >> 8) ^ (uint32_t)(i & 1) : ((uint32_t)(i / 64) * 2654435761u >> 8) + (uint32_t)(i & 7);
        }
        hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemset(d_tab, 0, (size_t)8 * table_words * 4);
        const uint32_t blocks = (n_per_table + 255) / 256 * 8;
        for (int pin = 1; pin >= 0; pin--)
            for (int mode = 0; mode < 5; mode++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
#define L(M, P) k<M, P><<<blocks, 256>>>(d_idx, n_per_table, d_tab, table_words, d_sink)
                    if (pin) { if (mode == 0) L(0, true); else if (mode == 1) L(1, true); else if (mode == 2) L(2, true); else if (mode == 3) L(3, true); else L(4, true); }
                    else { if (mode == 0) L(0, false); else if (mode == 1) L(1, false); else if (mode == 2) L(2, false); else if (mode == 3) L(3, false); else L(4, false); }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const char* mn[5] = {"pk_add_f16", "pk_add_f16 sc1", "add_f32", "add_f32 sc1", "plain load"};
                printf("pattern %d (%s) %s %-15s: %7.1f us for 8 tables x %u ops -> %6.1f G ops/s chip-wide\n", pattern,
                       pattern == 0 ? "random" : pattern == 1 ? "x pairs" : "runs of 8", pin ? "table pinned to XCD" : "tables on any XCD  ", mn[mode],
                       best * 1e3, n_per_table, 8.0 * n_per_table / (best * 1e-3) * 1e-9);
            }
    }
    // correctness of the scope-less form under pinning: every add must land (sum of a table == number of ops on it)
    hipMemset(d_tab, 0, (size_t)8 * table_words * 4);
    const uint32_t blocks = (n_per_table + 255) / 256 * 8;
    k<2, true><<<blocks, 256>>>(d_idx, n_per_table, d_tab, table_words, d_sink);
    hipDeviceSynchronize();
    std::vector<float> t((size_t)8 * table_words);
    hipMemcpy(t.data(), d_tab, t.size() * 4, hipMemcpyDeviceToHost);
    for (int tb = 0; tb < 8; tb++) { double s = 0; for (uint32_t i = 0; i < table_words; i++) s += t[(size_t)tb * table_words + i]; printf("table %d: sum %.0f (expected %u)\n", tb, s, n_per_table); }
    return 0;
}
