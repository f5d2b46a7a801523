// gridencoder.hip -- multiresolution hash / tiled grid encoder for gfx950.
//
// Replaces gridencoder/src/gridencoder.cu of the reference (cited per kernel).
//
// MI355X mapping: one lane = one (sample, level).  Blocks are dealt round-robin
// over the 8 XCDs (block b and b+8 share an XCD, MI355X_MICROARCH "Workgroup
// dispatch"), and each XCD has a private 4 MiB L2.  When L is a multiple of 8 the
// blockIdx -> (level, chunk) map sends all blocks of level l to XCD (l mod 8), one
// level at a time, so a hashed level's table (2 MiB fp16 / 4 MiB fp32 at T=2^19)
// is gathered out of ONE L2 instead of being replicated into all eight.
//
// Per-level scale = exp2(level*S)*H - 1 is evaluated once on the host (the
// reference evaluates exp2f per thread, gridencoder.cu:138) and passed by value;
// the oracle uses the same host libm, so fp32 results are bit-identical to it.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "lae_common.h"
#include <vector>
#include <mutex>
#include <utility>

namespace {

constexpr int MAX_LEVELS = 32;
// scale[l] = exp2(l*S)*H - 1 (host, :138).  in_shift / in_scale: optional affine map applied to every coordinate as it
// is read, x01 = (x + in_shift) * in_scale -- GridEncoder.forward's `(inputs + bound) / (2 * bound)` (grid.py:149; torch
// evaluates the division by a Python scalar as a multiplication by its fp32 reciprocal) without two extra kernels.
struct LevelScales { float scale[MAX_LEVELS]; float in_shift, in_scale; };

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__constant__ uint32_t k_primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};

// in-kernel phase stamps of the binned backward for tools/grid_bwd_stamps.py (compiled in only with -DLAE_GRID_STAMPS: the
// 100 MHz wall clock as thread 0 of a block passes each phase; slot = block, 32 stamps each)
#ifdef LAE_GRID_STAMPS
__device__ unsigned long long g_grid_stamps[32768 * 32];     // (8 MB, probe builds only: a 1080p frame's encoder launch is 216 k blocks)
#define GRID_STAMP(slot, i) do { if (threadIdx.x == 0 && (uint32_t)(i) < 32u) g_grid_stamps[(size_t)(slot) * 32 + (i)] = wall_clock64(); } while (0)
#define GRID_NOTE(slot, i, v) do { if (threadIdx.x == 0 && (uint32_t)(i) < 32u) g_grid_stamps[(size_t)(slot) * 32 + (i)] = (v); } while (0)
#else
#define GRID_STAMP(slot, i) do { } while (0)
#define GRID_NOTE(slot, i, v) do { } while (0)
#endif


// block-uniform description of one level (all derived from scalars -> SGPRs)
template <int D>
struct LevelInfo {
    float scale;
    uint32_t resolution, hashmap_size, table_off;
    uint32_t stride[D];      // dense strides of the dims that fit (gridencoder.cu:72-75)
    uint32_t ndense;         // how many dims the dense loop consumed
    bool use_hash, pow2;
    bool nowrap;             // dense index provably < hashmap_size (all dims indexed and (res+1)^D <= size): no modulo needed
};

template <int D>
__device__ __forceinline__ LevelInfo<D> level_info(const LevelScales& sc, const int32_t* __restrict__ offsets,
                                                   uint32_t level, uint32_t gridtype, bool align_corners) {
    LevelInfo<D> li;
    li.scale = sc.scale[level];
    li.resolution = (uint32_t)ceilf(li.scale) + 1;
    li.table_off = (uint32_t)offsets[level];
    li.hashmap_size = (uint32_t)offsets[level + 1] - li.table_off;
    uint32_t stride = 1;
    li.ndense = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        li.stride[d] = 0;
        if (li.ndense == (uint32_t)d && stride <= li.hashmap_size) {
            li.stride[d] = stride;
            stride *= align_corners ? li.resolution : (li.resolution + 1);
            li.ndense = d + 1;
        }
    }
    li.use_hash = (gridtype == 0) && (stride > li.hashmap_size);
    li.nowrap = (li.ndense == (uint32_t)D) && (stride <= li.hashmap_size);
    li.pow2 = (li.hashmap_size & (li.hashmap_size - 1)) == 0;
    return li;
}

// gridencoder.cu:66-84
template <int D>
__device__ __forceinline__ uint32_t cell_index(const LevelInfo<D>& li, const uint32_t (&pg)[D]) {
    uint32_t index = 0;
    if (li.use_hash) {
#pragma unroll
        for (int d = 0; d < D; d++) index ^= pg[d] * k_primes[d];
    } else {
#pragma unroll
        for (int d = 0; d < D; d++) index += pg[d] * li.stride[d];   // stride 0 for dims the loop skipped
    }
    return li.pow2 ? (index & (li.hashmap_size - 1)) : (index % li.hashmap_size);
}

__device__ __forceinline__ void block_to_level_chunk(uint32_t nb, bool xcd_mode, uint32_t& level, uint32_t& chunk) {
    const uint32_t bid = blockIdx.x;
    if (xcd_mode) {
        const uint32_t xcd = bid & 7u, j = bid >> 3;
        level = xcd + 8u * (j / nb);
        chunk = j % nb;
    } else {
        level = bid / nb;
        chunk = bid % nb;
    }
}

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

// accumulate r += w * v with the reference's rounding (`results[ch] += w * grid[..]`, :187): fp32 -> one FMA (nvcc
// contraction).  scalar_t = at::Half: `float * Half` is a float, and the only viable `Half += float` is
// operator+=(Half&, const Half&) -- the product is ROUNDED TO HALF first, then the two halves are added (through float,
// rounded once: exactly the correctly rounded half sum).  Checked against torch's own Half header: 0 mismatches in 2e6
// random cases, 15 % mismatches for a model that keeps the product in fp32.
__device__ __forceinline__ void accum(float& r, float w, float v) { r = fmaf(w, v, r); }
__device__ __forceinline__ void accum(half_t& r, float w, half_t v) { r = r + (half_t)(w * (float)v); }

constexpr int GRID_BLOCK = 256;

// ---------------------------------------------------------------- K13
// gridencoder.cu:87-245.  out index = b*os_b + level*os_l + ch  ([L,B,C]: os_b=C, os_l=B*C; [B,L*C]: os_b=L*C, os_l=C)
template <typename T, int D, int C>
__global__ __launch_bounds__(GRID_BLOCK) void k_grid_fwd(
    const float* __restrict__ inputs, const T* __restrict__ grid, const int32_t* __restrict__ offsets,
    T* __restrict__ outputs, uint32_t B, uint32_t L, LevelScales sc, T* __restrict__ dy_dx, uint32_t gridtype,
    bool align_corners, uint32_t interp, uint32_t nb, bool xcd_mode, uint64_t os_b, uint64_t os_l,
    const uint32_t* __restrict__ B_dev) {
    uint32_t level, chunk;
    block_to_level_chunk(nb, xcd_mode, level, chunk);
    if (level >= L) return;
    const uint32_t b = chunk * GRID_BLOCK + threadIdx.x;
    // B_dev (frame loop): the live row count is produced on the device; B stays the capacity the strides are built from
    if (B_dev) B = min(B, *B_dev);
    if (b >= B) return;
    const LevelInfo<D> li = level_info<D>(sc, offsets, level, gridtype, align_corners);
    const T* __restrict__ tab = grid + (size_t)li.table_off * C;

    float x[D];
    bool oob = false;
#pragma unroll
    for (int d = 0; d < D; d++) {
        x[d] = (inputs[(size_t)b * D + d] + sc.in_shift) * sc.in_scale;
        oob |= (x[d] < 0.0f) | (x[d] > 1.0f);
    }
    T* out = outputs + (size_t)b * os_b + (size_t)level * os_l;
    T* dout = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : nullptr;
    if (oob) {                                             // :118-135
#pragma unroll
        for (int ch = 0; ch < C; ch++) out[ch] = (T)0.0f;
        if (dout) {
#pragma unroll
            for (int i = 0; i < D * C; i++) dout[i] = (T)0.0f;
        }
        return;
    }

    float frac[D], dfrac[D];
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; d++) {                          // :146-159
        float p = fmaf(x[d], li.scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)fl;
        p -= (float)pg[d];
        if (interp == 1) { dfrac[d] = 6 * p * (1.0f - p); p = p * p * (3.0f - 2.0f * p); }
        else dfrac[d] = 1.0f;
        frac[d] = p;
    }

    T res[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) res[ch] = (T)0.0f;
    if constexpr (D == 3 && C == 2) {
        // Corner pairs (x, x+1): the two entries are adjacent and pair-aligned whenever idx(x+1) == idx(x) ^ 1 -- on
        // hashed levels for every even x ((x ^ h) and ((x|1) ^ h) differ in bit 0 only), on dense levels for even
        // idx -- and are then fetched with ONE 8/16-byte load.  The kernel is bound by the number of cache lines the
        // texture path touches (~1 line / cycle / CU, profiles/r1_grid_fwd_pmc.txt); this removes ~25 % of them.
        // Accumulation order is unchanged (idx = x + 2y + 4z), so results stay bit-identical.
        using V2 = typename std::conditional<sizeof(T) == 2, half2_t, float2>::type;
        const V2* __restrict__ tab2 = reinterpret_cast<const V2*>(tab);
        T cv[8][2];
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            uint32_t pl[3] = {pg[0], pg[1] + (yz & 1), pg[2] + (yz >> 1)};
            const uint32_t i0 = cell_index<D>(li, pl);
            pl[0] = pg[0] + 1;
            const uint32_t i1 = cell_index<D>(li, pl);
            V2 a, b;
            if (i1 == (i0 ^ 1u)) {
                if constexpr (sizeof(T) == 2) {
                    const uint2 w = *reinterpret_cast<const uint2*>(tab2 + (i0 & ~1u));
                    const half2_t lo = __builtin_bit_cast(half2_t, w.x), hi = __builtin_bit_cast(half2_t, w.y);
                    a = (i0 & 1u) ? hi : lo; b = (i0 & 1u) ? lo : hi;
                } else {
                    const float4 w = *reinterpret_cast<const float4*>(tab2 + (i0 & ~1u));
                    a = (i0 & 1u) ? make_float2(w.z, w.w) : make_float2(w.x, w.y);
                    b = (i0 & 1u) ? make_float2(w.x, w.y) : make_float2(w.z, w.w);
                }
            } else { a = tab2[i0]; b = tab2[i1]; }
            if constexpr (sizeof(T) == 2) { cv[2 * yz][0] = a[0]; cv[2 * yz][1] = a[1]; cv[2 * yz + 1][0] = b[0]; cv[2 * yz + 1][1] = b[1]; }
            else { cv[2 * yz][0] = a.x; cv[2 * yz][1] = a.y; cv[2 * yz + 1][0] = b.x; cv[2 * yz + 1][1] = b.y; }
        }
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const float w = (((idx & 1) ? frac[0] : 1 - frac[0]) * ((idx & 2) ? frac[1] : 1 - frac[1])) * ((idx & 4) ? frac[2] : 1 - frac[2]);
            accum(res[0], w, cv[idx][0]);
            accum(res[1], w, cv[idx][1]);
        }
    } else {
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {             // :166-191
        float w = 1.0f;
        uint32_t pgl[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            if ((idx & (1 << d)) == 0) { w *= 1 - frac[d]; pgl[d] = pg[d]; }
            else { w *= frac[d]; pgl[d] = pg[d] + 1; }
        }
        const uint32_t gi = cell_index<D>(li, pgl) * C;
        T v[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) v[ch] = tab[gi + ch];
#pragma unroll
        for (int ch = 0; ch < C; ch++) accum(res[ch], w, v[ch]);
    }
    }
    if constexpr (C == 2 && sizeof(T) == 2) {
        half2_t h = {res[0], res[1]};
        *reinterpret_cast<half2_t*>(out) = h;
    } else if constexpr (C == 2 && sizeof(T) == 4) {
        *reinterpret_cast<float2*>(out) = make_float2(res[0], res[1]);
    } else {
#pragma unroll
        for (int ch = 0; ch < C; ch++) out[ch] = res[ch];
    }

    if (dout) {                                            // :201-243
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
            T rg[C];
#pragma unroll
            for (int ch = 0; ch < C; ch++) rg[ch] = (T)0.0f;
#pragma unroll
            for (int idx = 0; idx < (1 << (D - 1)); idx++) {
                float w = li.scale;
                uint32_t pgl[D];
#pragma unroll
                for (int nd = 0; nd < D - 1; nd++) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((idx & (1 << nd)) == 0) { w *= 1 - frac[d]; pgl[d] = pg[d]; }
                    else { w *= frac[d]; pgl[d] = pg[d] + 1; }
                }
                pgl[gd] = pg[gd];
                const uint32_t il = cell_index<D>(li, pgl) * C;
                pgl[gd] = pg[gd] + 1;
                const uint32_t ir = cell_index<D>(li, pgl) * C;
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    if constexpr (sizeof(T) == 2) {
                        const half_t diff = (half_t)((float)tab[ir + ch] - (float)tab[il + ch]);
                        rg[ch] = rg[ch] + (half_t)(w * (float)diff * dfrac[gd]);     // Half += float: see accum()
                    } else {
                        rg[ch] = fmaf(w * (tab[ir + ch] - tab[il + ch]), dfrac[gd], rg[ch]);   // `+=` of a product: nvcc contracts (:236)
                    }
                }
            }
#pragma unroll
            for (int ch = 0; ch < C; ch++) dout[gd * C + ch] = rg[ch];
        }
    }
}

// ---------------------------------------------------------------- K13, hot configuration
// fp16 table, D = 3, C = 2, linear interpolation, align_corners = false, hash grid type, no dy_dx: every shipped config
// (grid.py:96-161 defaults under autocast).  Same arithmetic per sample as k_grid_fwd (bit-identical results), but
//  * the three level kinds are separate, branch-free code paths selected by a block-uniform switch:
//    hashed with a power-of-two size -- the y / z hash terms cost one multiply + one add each instead of eight
//    multiplies, byte offsets are formed before the xor (no per-corner shift), whether (x, x+1) share an aligned
//    8-byte pair is decided once per lane (x even) instead of once per corner row, 32-bit offsets against a uniform
//    base; dense without wrap-around -- eight plain loads off one base index (the divergent pair path cost more than
//    it saved: an unaligned 8-byte load is issued as two), no modulo; anything else -- the generic index.
//    110 VALU instructions per (sample, level) instead of ~200: measured alone on one XCD (tools/ubench/
//    grid_fwd_variants.hip, 433 k samples) a dense level takes 15.5 us instead of 29, level 11 44 instead of 57.
//  * work is dealt to the XCDs by a host-built schedule (FwdSched) instead of the fixed (l, l + 8) pairing: a level's
//    cost is set by its vector-memory instruction issue (coarse levels) or by the L2 request rate of its XCD (fine
//    levels, one 64-byte sector per (y, z) corner row and sample: ~57 us per level and 433 k samples whatever the
//    kernel does), so whole hashed levels are packed longest-first and the cheap dense levels (tables <= 0.8 MB, harmless
//    to replicate in several L2s) fill the gaps in eighths.
struct FwdSeg { uint32_t level, c0, n, cnt; };             // chunks [c0, c0 + n) of `level`; cnt > 0: the n chunks (j / cnt) * 8 + c0 + j % cnt instead (residues c0 .. c0 + cnt - 1 mod 8)
constexpr int FWD_MAX_SEG = 12;
struct FwdSched { uint32_t nseg[8]; FwdSeg seg[8][FWD_MAX_SEG]; };

// one (sample row, level) of the lean forward: shared by k_grid_fwd_lean and by the frame loop's overflow launch below
__device__ __forceinline__ void grid_fwd_lean_row(const float* __restrict__ inputs, const half_t* __restrict__ grid, const int32_t* __restrict__ offsets,
                                                  half_t* __restrict__ outputs, const LevelScales& sc, uint64_t os_b, uint64_t os_l, uint32_t level, uint32_t b) {
#include "grid_fwd_lean_row.inc"
}

__global__ __launch_bounds__(GRID_BLOCK) void k_grid_fwd_lean(
    const float* __restrict__ inputs, const half_t* __restrict__ grid, const int32_t* __restrict__ offsets,
    half_t* __restrict__ outputs, uint32_t B, LevelScales sc, FwdSched sched, uint64_t os_b, uint64_t os_l,
    const uint32_t* __restrict__ B_dev
#ifdef LAE_GRID_FWD_LOOP_PROBE
    , uint32_t pass_chunks                                 // fault probe only (below): 0 = one trip
#endif
    ) {
    const uint32_t xcd = blockIdx.x & 7u;
#ifdef LAE_GRID_STAMPS
    const unsigned long long st_t0 = wall_clock64();
#endif
    uint32_t j = blockIdx.x >> 3, level = 0xffffffffu, chunk = 0;
    const uint32_t ns = sched.nseg[xcd];
    for (uint32_t q = 0; q < ns; q++) {
        const uint32_t n = sched.seg[xcd][q].n;
        if (j < n) {
            const uint32_t cnt = sched.seg[xcd][q].cnt;
            level = sched.seg[xcd][q].level;
            chunk = cnt ? (j / cnt) * 8u + sched.seg[xcd][q].c0 + j % cnt : sched.seg[xcd][q].c0 + j;
            break;
        }
        j -= n;
    }
    if (level == 0xffffffffu) return;
#ifndef LAE_GRID_FWD_LOOP_PROBE
    const uint32_t b = chunk * GRID_BLOCK + threadIdx.x;
    if (B_dev) B = min(B, *B_dev);
    if (b >= B) return;
    grid_fwd_lean_row(inputs, grid, offsets, outputs, sc, os_b, os_l, level, b);
#else
    // FAULT PROBE, never in the shipped library (tools/grid_loop_fault.sh builds it into tools/ubench/bin/): round 4's reverted
    // "workgroups loop over the overflow rows" form of this kernel (commit 4541188).  With the row statements inside a
    // by-reference lambda inside this loop (variant 1) a few hundred rays of a frame differed from run to run as soon as a SECOND
    // PROCESS rendered frames on the GPU -- also with one trip (pass_chunks = 0, what the probe launches); the same loop around the
    // __forceinline__ function (variant 2) never did.  Round 5 bisected the two listings (they differ in the scalar prologue's
    // schedule and in ONE vector instruction) by hand-patched assembly: the instruction is
    //       v_pk_mul_f32 v[20:21], v[6:7], v[22:23] op_sel:[0,1] op_sel_hi:[0,1]          (variant 2: v[22:23], v[6:7] op_sel:[1,0] op_sel_hi:[1,0])
    // i.e. the gfx950 packed-fp32 erratum of laenerf_amd/build.py: beside the other process's MFMA kernels the low product is
    // sometimes a.lo x 0.  Built with the compiler's defaults (--packed-fp32) the probe still shows it; built like the shipped
    // library (no packed-fp32 instructions) it is gone.
    if (B_dev) B = min(B, *B_dev);
    for (uint32_t b = chunk * GRID_BLOCK + threadIdx.x; b < B; b += pass_chunks * GRID_BLOCK) {
#if LAE_GRID_FWD_LOOP_PROBE == 2
        grid_fwd_lean_row(inputs, grid, offsets, outputs, sc, os_b, os_l, level, b);
#else
        [&]() {
#include "grid_fwd_lean_row.inc"
        }();
#endif
        if (pass_chunks == 0u) break;
    }
#endif
#ifdef LAE_GRID_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 32768u * 8u) {            // (same-address atomics per XCD stretched the launch from 50 to 290 us: plain stores, a slot per block)
        g_grid_stamps[(size_t)blockIdx.x * 4] = st_t0; g_grid_stamps[(size_t)blockIdx.x * 4 + 1] = wall_clock64(); g_grid_stamps[(size_t)blockIdx.x * 4 + 2] = level;
    }
#endif
}


// Frame loop, rows [b0, *B_dev): the rows beyond what the host EXPECTED when it sized the lean launch (its lagging bound of the rays
// alive x the n_step that bound implies).  A launch sized for the worst case -- the row budget -- had 130 k of its 216 k workgroups
// find no rows in the iterations of a 1080p frame (~0.3 ns each: 40 of 245 us).  Rows exist here only in the few iterations after
// n_step rose on the device; then one workgroup takes 256 rows through ALL levels (straight-line code, one copy per level: no
// XCD placement for these rows, and no loop around the row code -- see DESIGN.md section 8).
__global__ __launch_bounds__(GRID_BLOCK) void k_grid_fwd_lean_tail(
    const float* __restrict__ inputs, const half_t* __restrict__ grid, const int32_t* __restrict__ offsets,
    half_t* __restrict__ outputs, uint32_t B, LevelScales sc, uint32_t L, uint64_t os_b, uint64_t os_l,
    const uint32_t* __restrict__ B_dev, uint32_t b0) {
    const uint32_t b = b0 + blockIdx.x * GRID_BLOCK + threadIdx.x;
    B = min(B, *B_dev);
    if (b >= B) return;
#pragma unroll
    for (uint32_t level = 0; level < 16u; level++)
        if (level < L) grid_fwd_lean_row(inputs, grid, offsets, outputs, sc, os_b, os_l, level, b);
}
// ---------------------------------------------------------------- K14
// gridencoder.cu:248-340: scatter w*grad into grad_grid.  v1: one no-return atomic per
// (corner, channel pair): global_atomic_add_f32 / global_atomic_pk_add_f16.
// ---- constants and predicates of the binned backward (defined here because the generic kernel below is its companion)
constexpr uint32_t PART_SHIFT = 12, PART = 1u << PART_SHIFT;   // table entries per partition (= per accumulate workgroup)
constexpr uint32_t BK_MAX = 512;              // partitions per level the binned path handles (level size <= 2^21 entries)
constexpr uint32_t BK_TARGET = 32;            // target workgroups per level (levels with fewer partitions are split into sub-ranges).  Round 4: 16 -> 32
                                              // (same-box A/B, 16 / 32 / 64: lego step 0.3514 / 0.3508 / 0.3547 ms, style step 0.396 / 0.367 / 0.368, flower step 0.700 / 0.666 / 0.647:
                                              // the smallest levels' few cells take every sample's LDS atomics, more sub-ranges spread them over more CUs)

struct LevelBins { uint32_t P, SUB; };
__host__ __device__ __forceinline__ LevelBins level_bins_of(uint32_t hashmap_size, uint32_t bk_target = BK_TARGET) {
    LevelBins lb;
    lb.P = (hashmap_size + PART - 1) >> PART_SHIFT;
    lb.SUB = lb.P >= bk_target ? 1u : (bk_target + lb.P - 1) / lb.P;
    return lb;
}
// A level goes through the binned pipeline when its partitions fit the directory rows; anything larger is left to the
// generic atomic kernel.
__host__ __device__ __forceinline__ bool level_is_binned(uint32_t hashmap_size) { return level_bins_of(hashmap_size).P <= BK_MAX; }


template <typename T, int D, int C>
__global__ __launch_bounds__(GRID_BLOCK) void k_grid_bwd(
    const T* __restrict__ grad, const float* __restrict__ inputs, const int32_t* __restrict__ offsets,
    T* __restrict__ grad_grid, uint32_t B, uint32_t L, LevelScales sc, uint32_t gridtype, bool align_corners,
    uint32_t interp, uint32_t nb, bool xcd_mode, uint64_t gs_b, uint64_t gs_l, bool only_unbinned) {
    uint32_t level, chunk;
    block_to_level_chunk(nb, xcd_mode, level, chunk);
    if (level >= L) return;
    const uint32_t b = chunk * GRID_BLOCK + threadIdx.x;
    if (b >= B) return;
    const LevelInfo<D> li = level_info<D>(sc, offsets, level, gridtype, align_corners);
    if (only_unbinned) {      // companion of the binned fast path: only the levels it leaves out
        if constexpr (D == 3) { if (level_is_binned(li.hashmap_size)) return; }
    }
    T* __restrict__ tab = grad_grid + (size_t)li.table_off * C;

    float frac[D];
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float xv = (inputs[(size_t)b * D + d] + sc.in_shift) * sc.in_scale;
        if (xv < 0.0f || xv > 1.0f) return;                // :276-281
        float p = fmaf(xv, li.scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)fl;
        p -= (float)pg[d];
        if (interp == 1) p = p * p * (3.0f - 2.0f * p);
        frac[d] = p;
    }
    const T* g = grad + (size_t)b * gs_b + (size_t)level * gs_l;
    float gc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) gc[ch] = (float)g[ch];

#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float w = 1.0f;
        uint32_t pgl[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            if ((idx & (1 << d)) == 0) { w *= 1 - frac[d]; pgl[d] = pg[d]; }
            else { w *= frac[d]; pgl[d] = pg[d] + 1; }
        }
        const uint32_t gi = cell_index<D>(li, pgl) * C;
        if constexpr (sizeof(T) == 2) {
            static_assert(C % 2 == 0 || sizeof(T) == 4, "fp16 grads need an even channel count");
#pragma unroll
            for (int ch = 0; ch < C; ch += 2) {
                half2_t v = {(half_t)(w * gc[ch]), (half_t)(w * gc[ch + 1])};      // :329
                __builtin_amdgcn_global_atomic_fadd_v2f16(
                    (__attribute__((address_space(1))) half2_t*)(tab + gi + ch), v);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ch++) atomicAdd(tab + gi + ch, w * gc[ch]);     // :336
        }
    }
}

// ---------------------------------------------------------------- K14, MI355X form (D = 3, C = 2), round 2
// Scattered global float atomics execute at the memory side on gfx950 (~20 G requests/s chip-wide, MI355X_MICROARCH
// "Global float atomics"): 128 of them per sample made k_grid_bwd 54 % of the first train step.  Instead the gradient
// table is cut into partitions of PART = 4096 entries and the contributions are ROUTED to the partition's owner by a
// counting sort:
//
//   k_bwd_walk<COUNT> : block = UNIT = (level, 1024 consecutive samples), 4 consecutive samples per lane.  While the cell
//                is unchanged the 8 corner contributions are summed in registers; every finished cell yields ITEMS: one
//                per (y', z') corner row = {key: entry e0 inside the partition | how to get e1 from e0, values of the x
//                and x+1 corners} (10 bytes for fp16 gradients) -- both x corners share a partition unless the pair
//                straddles a partition boundary, which costs two single-corner items.  COUNT only needs the positions:
//                items per partition (LDS histogram) -> the unit's counter row + its column of the per-partition table.
//   k_bwd_scan_units / k_bwd_scan_parts : exclusive scans over the units of a partition and over the partitions:
//                where every unit's run of every partition starts in the queue.
//   k_bwd_walk<FILL>  : same walk with the gradients; items are ranked inside the block (LDS counters), laid out
//                partition by partition in an LDS staging area and leave for the queue as whole runs (consecutive lanes,
//                consecutive slots: in round 1 every 16-byte item was its own L2 request and the pass ran at the L2
//                request rate).  No global atomics anywhere.
//   k_bwd_acc  : one workgroup per (level, partition[, sub-range]) -- handed out through a device-side ticket, fullest
//                first -- streams its contiguous queue range, accumulates in LDS and adds the partition to the table with
//                coalesced read-modify-writes.  fp16 gradients accumulate EXACTLY as 2^-24 fixed point in int64 LDS words
//                (every fp16 value is a multiple of 2^-24; random-address ds_add_u64 costs ~0.3 cycles per lane-op,
//                tools/ubench/lds_acc2.hip), and the table entry becomes RN_half(old + sum) with ONE rounding: order
//                independent, deterministic.  A non-finite contribution (overflowed loss scale) marks its entry, which is
//                written as NaN so that the optimizer's non-finite scan sees it (an integer accumulator would otherwise
//                silently turn Inf into a finite number).  fp32 gradients use ds_add_f32.
// Levels with few partitions are split into SUB sub-ranges so that ~16 workgroups share every level; each adds its exact
// partial sums (int64: the adds commute) into the partition's merge record with device-scope atomics and the last to arrive
// takes them out (no float atomics: same bits whatever the order).
// Levels with more than BK_MAX partitions (T > 2^21) are left to k_grid_bwd.
template <typename T> struct BVal;
template <> struct BVal<half_t> { using type = uint2; };       // {half2 of corner x, half2 of corner x+1}
template <> struct BVal<float> { using type = float4; };
constexpr int FILL_THREADS = 256;
constexpr int SPT = 4;                                         // consecutive samples per lane and segment
constexpr int SEGS = 1;                                        // segments per block: a lane walks SPT * SEGS consecutive samples
constexpr uint32_t UNIT_SAMPLES = FILL_THREADS * SPT;          // 512
constexpr uint32_t STAGE_CAP = 4096 + 256;                     // items staged in LDS: a fine level's unit is 1024 cells x 4 pair rows = 4096 items
                                                               // PLUS 4 for every cell whose x-neighbour lies in the next partition (1 cell in 4096);
                                                               // with a cap of exactly 4096, 22 % of the fine levels' units took the unstaged path
constexpr uint32_t KEY_SINGLE = 15u;                           // key code: 0..11 -> e1 = e0 ^ ((2 << code) - 1) (e0 + 1 is such an xor too: the trailing ones of e0 and the zero above them flip)
constexpr int ACC_THREADS = 1024;
constexpr uint32_t BWD_MAX_SAMPLES = 1u << 24;                  // byte offsets of the buffer loads (walk: 12 B per sample; accumulate: 8 B per item, 8 items per sample and level) stay below 2^31
constexpr uint32_t COARSE_RES = 64;                            // levels coarser than this: consecutive queue items often repeat an entry (same ray, same cell)
constexpr bool WIDE_COARSE_UNITS = false;                      // true: 16 samples per lane on those levels (round 3 experiment: accumulate 62.4 -> 58.4 us, but fill 58.3 -> 80.1 and count 27.5 -> 38.0 -- the wide blocks walk four dependent load groups and form the pass's tail; step 0.356 -> 0.433 ms)
constexpr uint32_t SUB_RECS = 64;                              // merge records per level: one per partition of a level that is split (P < BK_TARGET)
constexpr uint32_t TICKET_ARRIVALS = 2;                        // tickets[0] = work queue; [2 + level * SUB_RECS + p] = arrivals
constexpr uint32_t TICKET_WORDS = TICKET_ARRIVALS + MAX_LEVELS * SUB_RECS;
template <typename T> constexpr uint32_t sub_rec_words() { return (sizeof(T) == 2 ? 2 * PART : PART) + PART / 64; }

// the plan: everything the fill / accumulate passes need to know about where items go (positions only)
struct BwdPlan {
    uint16_t* cnt;        // [L * U][BK_MAX]   per unit: items per (partition, lane copy) counter
    uint32_t* part_cnt;   // [L][BK_MAX][U]    per partition: items of every unit, then (in place) their exclusive prefix
    uint32_t* totals;     // [L][BK_MAX]       items per partition
    uint32_t* offs;       // [L][BK_MAX]       first queue slot of every partition
    uint32_t* tickets;    // [TICKET_WORDS]
};

__device__ __forceinline__ uint32_t pack_half2(float a, float b) {
    const half2_t h = {(half_t)a, (half_t)b};
    return __builtin_bit_cast(uint32_t, h);
}

// PLAIN: linear interpolation, align_corners = false, hash grid type (every shipped config) as compile-time facts -- the pass
// is bound by its instruction count
template <typename T, bool FILL, bool PLAIN>
__global__ __launch_bounds__(FILL_THREADS) void k_bwd_walk(
    const T* __restrict__ gradT, const float* __restrict__ inputs, const int32_t* __restrict__ offsets, uint32_t B,
    uint32_t L, LevelScales sc, uint32_t gridtype_, bool align_corners_, uint32_t interp_, uint32_t U, BwdPlan plan,
    typename BVal<T>::type* __restrict__ qvals, uint16_t* __restrict__ qkeys, uint32_t* __restrict__ touched_,
    uint32_t* __restrict__ level_full) {
    const uint32_t gridtype = PLAIN ? 0u : gridtype_, interp = PLAIN ? 0u : interp_;
    const bool align_corners = PLAIN ? false : align_corners_;
    using V = typename BVal<T>::type;
    if constexpr (FILL) GRID_STAMP(blockIdx.x, 0);
    const uint32_t NBLK = U / SEGS;                           // U is a multiple of SEGS
    const uint32_t level = blockIdx.x / NBLK, chunk = blockIdx.x % NBLK;
    if (FILL && blockIdx.x == 0) for (uint32_t k = threadIdx.x; k < TICKET_WORDS; k += FILL_THREADS) plan.tickets[k] = 0;   // work queue / arrival counters of the accumulate pass
    const LevelInfo<3> li = level_info<3>(sc, offsets, level, gridtype, align_corners);
    const uint32_t P = (li.hashmap_size + PART - 1) >> PART_SHIFT;
    if (P > BK_MAX) return;                                // handled by the generic atomic kernel
    __shared__ uint32_t hist[BK_MAX], start[BK_MAX + 1], gbase[FILL ? BK_MAX : 1];
    __shared__ __attribute__((aligned(16))) V s_vals[FILL ? STAGE_CAP : 1];
    __shared__ __attribute__((aligned(16))) uint16_t s_keys[FILL ? STAGE_CAP : 8];
    extern __shared__ uint32_t s_lines[];                  // count pass with `touched`: line bits of this level, aligned to the global words
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // bit of entry e of this level in the caller's bitmap: (table_off + e) / 8; the LDS copy starts at the bitmap word that
    // holds the level's first line
    const uint32_t line_shift = (li.table_off >> 3) & 31u;
    const uint32_t line_words = (line_shift + ((li.hashmap_size + 7u) >> 3) + 31u) >> 5;
    // a level all of whose lines are marked has nothing to add: level_full[level] (set below by a block that sees it) switches
    // the marking of that level off for good (on the bench scene: levels 9-15 after a few steps)
    // The decision is made ONCE per block (thread 0 reads the flag, the block agrees through LDS): blocks of this very launch
    // set level_full[level], so a per-thread read could split a block's waves between marking and not marking -- mismatched
    // barrier counts below and uninitialised LDS words ORed into the bitmap.
    __shared__ uint32_t s_mark;
    if (!FILL) {
        if (tid == 0) s_mark = (touched_ && !level_full[level]) ? 1u : 0u;
        __syncthreads();
    }
    uint32_t* __restrict__ touched = (!FILL && s_mark) ? touched_ : nullptr;
    if (!FILL && touched) { for (uint32_t k = tid; k < line_words; k += FILL_THREADS) s_lines[k] = 0u; }
    // a lane walks SPT * SEGS consecutive samples; every segment of SPT samples per lane is a UNIT with its own counter
    // row / queue runs (the staging area holds one unit), the cell a lane is in is carried from segment to segment
    // Round 3: on the levels coarser than COARSE_RES a lane walks 16 consecutive samples (four groups of SPT), not four: a cell
    // of such a level holds ~10-40 consecutive samples of a ray, so the per-lane merge of same-cell samples yields up to four
    // times fewer items -- and those items' same-address LDS atomics are the chain that sets the accumulate pass's duration
    // (DESIGN.md 4d).  A unit of a coarse level is then 4096 samples; it lives in the unit slot of its first 1024 (slots
    // u % 4 != 0 stay empty: zero counters, nothing filled), so the plan's layout does not change.
    const bool wide = li.resolution < COARSE_RES && WIDE_COARSE_UNITS;
    const uint32_t NSUB = wide ? 4u : 1u;
    if (wide && (chunk & 3u)) {
        if constexpr (!FILL) {                                   // an empty unit slot: its counters must still read zero
            const uint32_t u_e = chunk * SEGS, unit_e = level * U + u_e;
            for (uint32_t k = tid; k < BK_MAX / 2; k += FILL_THREADS) reinterpret_cast<uint32_t*>(plan.cnt + (size_t)unit_e * BK_MAX)[k] = 0u;
            for (uint32_t k = tid; k < P; k += FILL_THREADS) plan.part_cnt[((size_t)level * BK_MAX + k) * U + u_e] = 0u;
        }
        return;
    }
    const uint32_t b0 = wide ? ((chunk >> 2) * (UNIT_SAMPLES * 4u) + tid * (SPT * 4u)) : (chunk * FILL_THREADS + tid) * (SPT * SEGS);
    // few partitions: all lanes of a wave count into the same one or two counters, and same-address LDS atomics of one
    // instruction serialise -> NC copies per partition, lane l uses copy l mod NC (sub-runs inside the partition's run).
    // Coarse levels keep arrival order instead: the accumulate pass merges neighbouring repeats.
    const uint32_t NC = (P <= BK_MAX / 8 && li.resolution >= COARSE_RES) ? 8u : 1u;
    const T* __restrict__ g_lvl = gradT + (size_t)level * B * 2;

    uint32_t cpg[3] = {0, 0, 0};
    bool have = false;
    float a0[8], a1[8];
#pragma unroll
    for (int c = 0; c < 8; c++) { a0[c] = 0.f; a1[c] = 0.f; }

    // items of the cell cp, one row of x-neighbours (y, z corner) at a time: a PAIR item {counter c0, key k0} when both
    // entries sit in one partition a low-bit xor apart, else two single items {c0, k0}, {c1, k1}
    struct Rows { uint32_t c0[4], k0[4], c1[4], k1[4], i0[4], i1[4]; bool pair[4]; };   // i0 / i1: the two entries of the row (count pass only)
    auto cell_rows = [&](const uint32_t (&cp)[3], Rows& r) {
        const uint32_t lane_copy = lane & (NC - 1);
        if (li.use_hash && li.pow2) {
            // hashed level with a power-of-two table: the x-neighbour is i1 = i0 ^ dx on EVERY row, dx = the low-bit mask
            // (x ^ (x + 1)) & (size - 1), so whether the row is a pair item and its key code are decided once per cell
            // (the pass is bound by its instruction count: ~30 instructions per row -> ~9)
            const uint32_t m = li.hashmap_size - 1u;
            const uint32_t dx = (cp[0] ^ (cp[0] + 1u)) & m;
            const bool pr = dx != 0u && dx < PART;
            const uint32_t code = (pr ? (uint32_t)(__builtin_popcount(dx) - 1) : KEY_SINGLE) << 12;
            const uint32_t hy0 = cp[1] * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = cp[2] * 805459861u, hz1 = hz0 + 805459861u;
#pragma unroll
            for (int yz = 0; yz < 4; yz++) {
                const uint32_t i0 = (cp[0] ^ ((yz & 1) ? hy1 : hy0) ^ ((yz & 2) ? hz1 : hz0)) & m, i1 = i0 ^ dx;
                r.c0[yz] = (i0 >> PART_SHIFT) * NC + lane_copy; r.c1[yz] = (i1 >> PART_SHIFT) * NC + lane_copy;
                r.i0[yz] = i0; r.i1[yz] = i1;
                r.pair[yz] = pr;
                r.k0[yz] = (i0 & (PART - 1)) | code;
                r.k1[yz] = (i1 & (PART - 1)) | (KEY_SINGLE << 12);
            }
            return;
        }
        uint32_t hy0 = 0, hy1 = 0, hz0 = 0, hz1 = 0, key0 = 0;
        if (li.use_hash) {
            hy0 = cp[1] * 2654435761u; hy1 = hy0 + 2654435761u;
            hz0 = cp[2] * 805459861u; hz1 = hz0 + 805459861u;
        } else key0 = cp[0] * li.stride[0] + cp[1] * li.stride[1] + cp[2] * li.stride[2];
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            uint32_t i0, i1;
            if (li.use_hash) {
                const uint32_t h = ((yz & 1) ? hy1 : hy0) ^ ((yz & 2) ? hz1 : hz0);
                i0 = cp[0] ^ h; i1 = (cp[0] + 1) ^ h;
            } else {
                i0 = key0 + ((yz & 1) ? li.stride[1] : 0u) + ((yz & 2) ? li.stride[2] : 0u);
                i1 = i0 + li.stride[0];
            }
            if (li.pow2) { i0 &= li.hashmap_size - 1; i1 &= li.hashmap_size - 1; }
            else if (li.use_hash || !li.nowrap) { i0 %= li.hashmap_size; i1 %= li.hashmap_size; }
            const uint32_t e0 = i0 & (PART - 1), e1 = i1 & (PART - 1), d = e0 ^ e1;
            const bool same_part = (i0 >> PART_SHIFT) == (i1 >> PART_SHIFT) && d != 0;
            r.c0[yz] = (i0 >> PART_SHIFT) * NC + lane_copy; r.c1[yz] = (i1 >> PART_SHIFT) * NC + lane_copy;
            r.i0[yz] = i0; r.i1[yz] = i1;
            r.pair[yz] = same_part && (d & (d + 1)) == 0;
            r.k0[yz] = e0 | ((r.pair[yz] ? (uint32_t)(__builtin_popcount(d) - 1) : KEY_SINGLE) << 12);
            r.k1[yz] = e1 | (KEY_SINGLE << 12);
        }
    };

    for (int seg = 0; seg < SEGS; seg++) {
        const uint32_t u = chunk * SEGS + seg, unit = level * U + u;
        const uint32_t bs = b0 + seg * SPT;
        const bool last = seg == SEGS - 1;
        // every global read of the unit is issued before the first one is consumed: one memory latency, not three
        static_assert(BK_MAX / 2 == FILL_THREADS && BK_MAX == 2 * FILL_THREADS, "one counter word and two partitions per lane");
        uint32_t w_cnt = 0, gb_off[2] = {0, 0}, gb_cnt[2] = {0, 0};
        if (!FILL) { for (uint32_t k = tid; k < BK_MAX; k += FILL_THREADS) hist[k] = 0; }
        else {
            // the unit's own counters (from the count pass) and where its run of every partition starts in the queue
            w_cnt = reinterpret_cast<const uint32_t*>(plan.cnt + (size_t)unit * BK_MAX)[tid];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t k = tid + i * FILL_THREADS;
                if (k < P) { gb_off[i] = plan.offs[level * BK_MAX + k]; gb_cnt[i] = plan.part_cnt[((size_t)level * BK_MAX + k) * U + u]; }
            }
        }
        // ---- positions (and gradients) of SPT consecutive samples: 16-byte buffer loads, the hardware range check
        // returns zeros past the end of the batch (no ragged-tail path for the compiler to blend into the wide one)
        static_assert(SPT == 4, "the wide reads below take 4 samples per lane");
        float xs[SPT][3], g0[SPT], g1[SPT];
        auto load_group = [&](uint32_t bs) {
            using u4 = __attribute__((__vector_size__(16))) uint32_t;
            // (the range check is all-or-nothing per 16-byte load: the records end with the last whole group of 4 samples,
            // the one lane that holds the batch's last 1-3 samples reads them with plain loads below)
            const uint32_t B4 = B & ~3u;
            const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inputs), 0, (int)(B4 * 12u), 0x00020000);
            const u4 x0 = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(bs * 12u), 0, 0);
            const u4 x1 = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(bs * 12u + 16u), 0, 0);
            const u4 x2 = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(bs * 12u + 32u), 0, 0);
            u4 ga = {0, 0, 0, 0}, gb = {0, 0, 0, 0};
            if constexpr (FILL) {
                const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(g_lvl), 0, (int)(B4 * 2u * (uint32_t)sizeof(T)), 0x00020000);
                ga = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)(bs * 2u * (uint32_t)sizeof(T)), 0, 0);
                if constexpr (sizeof(T) == 4) gb = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)(bs * 8u + 16u), 0, 0);
            }
            uint32_t t[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
            uint32_t gw[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
            if (bs == B4 && B4 != B) {
#pragma unroll
                for (int s_ = 0; s_ < 3; s_++) {
                    if (bs + s_ >= B) break;
#pragma unroll
                    for (int d = 0; d < 3; d++) t[3 * s_ + d] = __builtin_bit_cast(uint32_t, inputs[(size_t)(bs + s_) * 3 + d]);
                    if constexpr (FILL && sizeof(T) == 2) gw[s_] = reinterpret_cast<const uint32_t*>(g_lvl)[bs + s_];
                    else if constexpr (FILL) { gw[2 * s_] = reinterpret_cast<const uint32_t*>(g_lvl)[2 * (bs + s_)]; gw[2 * s_ + 1] = reinterpret_cast<const uint32_t*>(g_lvl)[2 * (bs + s_) + 1]; }
                }
            }
#pragma unroll
            for (int s_ = 0; s_ < SPT; s_++) {
#pragma unroll
                for (int d = 0; d < 3; d++) xs[s_][d] = __builtin_bit_cast(float, t[3 * s_ + d]);
                if constexpr (FILL && sizeof(T) == 2) { const half2_t h = __builtin_bit_cast(half2_t, gw[s_]); g0[s_] = (float)h[0]; g1[s_] = (float)h[1]; }
                else if constexpr (FILL) { g0[s_] = __builtin_bit_cast(float, gw[2 * s_]); g1[s_] = __builtin_bit_cast(float, gw[2 * s_ + 1]); }
                else { g0[s_] = 0.f; g1[s_] = 0.f; }
            }
        };
        load_group(bs);
        if constexpr (FILL) {
            hist[2 * tid] = w_cnt & 0xffffu; hist[2 * tid + 1] = w_cnt >> 16;
#pragma unroll
            for (int i = 0; i < 2; i++) { const uint32_t k = tid + i * FILL_THREADS; if (k < P) gbase[k] = gb_off[i] + gb_cnt[i]; }
        }
        uint32_t pgs[SPT][3]; float frs[SPT][3]; bool oks[SPT];
        auto place_group = [&](uint32_t bs) {
#pragma unroll
        for (int s_ = 0; s_ < SPT; s_++) {
            bool ok = bs + s_ < B;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float xv = (xs[s_][d] + sc.in_shift) * sc.in_scale;
                ok = ok && !(xv < 0.0f) && !(xv > 1.0f);                     // gridencoder.cu:276-281
                float pp = fmaf(xv, li.scale, align_corners ? 0.0f : 0.5f);
                const float fl = floorf(pp);
                pgs[s_][d] = (uint32_t)fl;
                pp -= (float)pgs[s_][d];
                if (interp == 1) pp = pp * pp * (3.0f - 2.0f * pp);
                frs[s_][d] = pp;
            }
            oks[s_] = ok;
        }
        };
        place_group(bs);
        if constexpr (FILL) GRID_STAMP(blockIdx.x, 1);
        __syncthreads();
        if constexpr (FILL) GRID_STAMP(blockIdx.x, 2);
        if constexpr (!FILL) {
            // ---- how many items does each (partition, copy) counter get from this unit
            auto count_cell = [&]() {
                Rows r;
                cell_rows(cpg, r);
#pragma unroll
                for (int yz = 0; yz < 4; yz++) atomicAdd(&hist[r.c0[yz]], 1u);
                if (touched) {
                    // "ever touched" bitmap of the caller, one bit per 8 entries (a 64-byte line of the fp32 table): the block
                    // notes the lines of its cells' corners in LDS (a superset of the lines that get a non-zero gradient is fine)
#pragma unroll
                    for (int yz = 0; yz < 4; yz++) {
                        const uint32_t l0 = line_shift + (r.i0[yz] >> 3), l1 = line_shift + (r.i1[yz] >> 3);
                        atomicOr(&s_lines[l0 >> 5], 1u << (l0 & 31u));
                        if (l1 != l0) atomicOr(&s_lines[l1 >> 5], 1u << (l1 & 31u));
                    }
                }
                if (!(r.pair[0] && r.pair[1] && r.pair[2] && r.pair[3])) {
#pragma unroll
                    for (int yz = 0; yz < 4; yz++) if (!r.pair[yz]) atomicAdd(&hist[r.c1[yz]], 1u);
                }
            };
            for (uint32_t sub = 0; sub < NSUB; sub++) {
                if (sub) { load_group(bs + sub * SPT); place_group(bs + sub * SPT); }
#pragma unroll
                for (int s_ = 0; s_ < SPT; s_++) {
                    if (!oks[s_]) continue;
                    if (!have || pgs[s_][0] != cpg[0] || pgs[s_][1] != cpg[1] || pgs[s_][2] != cpg[2]) {
                        if (have) count_cell();
                        cpg[0] = pgs[s_][0]; cpg[1] = pgs[s_][1]; cpg[2] = pgs[s_][2]; have = true;
                    }
                }
            }
            if (last && have) count_cell();
            __syncthreads();
            for (uint32_t k = tid; k < BK_MAX / 2; k += FILL_THREADS)
                reinterpret_cast<uint32_t*>(plan.cnt + (size_t)unit * BK_MAX)[k] = hist[2 * k] | (hist[2 * k + 1] << 16);   // <= 8192 each
            for (uint32_t k = tid; k < P; k += FILL_THREADS) {
                uint32_t n = 0;
                for (uint32_t c = 0; c < NC; c++) n += hist[k * NC + c];
                plan.part_cnt[((size_t)level * BK_MAX + k) * U + u] = n;
            }
            if (touched) {                                     // new line bits of this block -> the caller's bitmap (a plain read first)
                uint32_t* gw = touched + ((li.table_off >> 3) >> 5);
                const uint32_t n_lines = (li.hashmap_size + 7u) >> 3;
                bool full = true;
                for (uint32_t k = tid; k < line_words; k += FILL_THREADS) {
                    const uint32_t w = s_lines[k], cur = gw[k];
                    if (w && (cur & w) != w) atomicOr(&gw[k], w);
                    // bits of word k that belong to this level: [line_shift, line_shift + n_lines) in the level's bit space
                    const uint32_t lo_bit = k == 0 ? line_shift : 0u;
                    const uint32_t end = line_shift + n_lines - 32u * k;          // bits of this word below `end` are the level's
                    uint32_t mask = end >= 32u ? 0xffffffffu : ((1u << end) - 1u);
                    mask &= ~((1u << lo_bit) - 1u);
                    full = full && (((cur | w) & mask) == mask);
                }
                if (__syncthreads_and(full ? 1 : 0) && tid == 0) level_full[level] = 1u;
            }
            __syncthreads();                                   // hist is zeroed again by the next segment
        } else {
            // ---- exclusive scan over the counters (one wave, 8 counters per lane) -> where each sub-run starts in the staging area
            if (tid < 64) {
                uint32_t v[BK_MAX / 64], sum = 0;
#pragma unroll
                for (int k = 0; k < (int)(BK_MAX / 64); k++) { v[k] = hist[tid * (BK_MAX / 64) + k]; sum += v[k]; }
                uint32_t run = lae::wave_incl_scan(sum) - sum;
#pragma unroll
                for (int k = 0; k < (int)(BK_MAX / 64); k++) {     // hist becomes the running slot of its sub-run
                    start[tid * (BK_MAX / 64) + k] = run; hist[tid * (BK_MAX / 64) + k] = run; run += v[k];
                }
                if (tid == 63) start[BK_MAX] = run;
            }
            __syncthreads();
            GRID_STAMP(blockIdx.x, 3);
            const uint32_t total = start[BK_MAX];
            const bool staged = total <= STAGE_CAP;
            // ---- sums + emission, sorted by partition (slot = sub-run start + arrival order inside it)
            auto emit_cell = [&]() {
                Rows r;
                cell_rows(cpg, r);
                // the slots of all rows first (independent LDS atomics in flight together), then the writes
                uint32_t sl0[4], sl1[4];
#pragma unroll
                for (int yz = 0; yz < 4; yz++) sl0[yz] = atomicAdd(&hist[r.c0[yz]], 1u);
                // four pair rows is the common cell: one test instead of a divergent branch per row and phase
                const bool all_pairs = r.pair[0] && r.pair[1] && r.pair[2] && r.pair[3];
                if (!all_pairs) {
#pragma unroll
                    for (int yz = 0; yz < 4; yz++) sl1[yz] = r.pair[yz] ? 0u : atomicAdd(&hist[r.c1[yz]], 1u);
                }
                auto put = [&](uint32_t slot, uint32_t ci, uint32_t key, int c, bool pair) {
                    V val;
                    if constexpr (sizeof(T) == 2) { val.x = pack_half2(a0[c], a1[c]); val.y = pair ? pack_half2(a0[c + 1], a1[c + 1]) : 0u; }
                    else { val.x = a0[c]; val.y = a1[c]; val.z = pair ? a0[c + 1] : 0.f; val.w = pair ? a1[c + 1] : 0.f; }
                    if (staged) { s_vals[slot] = val; s_keys[slot] = (uint16_t)key; }
                    else {          // more items than the staging area holds (split corner pairs): straight to the queue
                        const uint32_t k = ci / NC, dst = gbase[k] + (slot - start[k * NC]);
                        qvals[dst] = val; qkeys[dst] = (uint16_t)key;
                    }
                };
                if (all_pairs) {
#pragma unroll
                    for (int yz = 0; yz < 4; yz++) put(sl0[yz], r.c0[yz], r.k0[yz], 2 * yz, true);
                } else {
#pragma unroll
                    for (int yz = 0; yz < 4; yz++) {
                        put(sl0[yz], r.c0[yz], r.k0[yz], 2 * yz, r.pair[yz]);
                        if (!r.pair[yz]) put(sl1[yz], r.c1[yz], r.k1[yz], 2 * yz + 1, false);
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; c++) { a0[c] = 0.f; a1[c] = 0.f; }
            };
            for (uint32_t sub = 0; sub < NSUB; sub++) {
                if (sub) { load_group(bs + sub * SPT); place_group(bs + sub * SPT); }
#pragma unroll
                for (int s_ = 0; s_ < SPT; s_++) {
                    GRID_STAMP(blockIdx.x, 8 + s_);
                    if (!oks[s_]) continue;
                    if (!have || pgs[s_][0] != cpg[0] || pgs[s_][1] != cpg[1] || pgs[s_][2] != cpg[2]) {
                        if (have) emit_cell();
                        cpg[0] = pgs[s_][0]; cpg[1] = pgs[s_][1]; cpg[2] = pgs[s_][2]; have = true;
                    }
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const float w = (((c & 1) ? frs[s_][0] : 1 - frs[s_][0]) * ((c & 2) ? frs[s_][1] : 1 - frs[s_][1])) * ((c & 4) ? frs[s_][2] : 1 - frs[s_][2]);
                        a0[c] = fmaf(w, g0[s_], a0[c]); a1[c] = fmaf(w, g1[s_], a1[c]);
                    }
                }
            }
            GRID_STAMP(blockIdx.x, 12);
            if (last && have) emit_cell();
            GRID_STAMP(blockIdx.x, 4);
            __syncthreads();
            GRID_STAMP(blockIdx.x, 5);
            if (staged) {
                // ---- the staged runs leave for the queue, one partition at a time per wave: consecutive lanes, consecutive slots
                // a wave owns a contiguous range of partitions; the (start, end, queue base) triples of up to 64 of them are
                // read by the lanes at once and handed round through SGPRs, so the copies of consecutive runs do not wait
                // for one another's LDS reads
                const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                const uint32_t per = (P + FILL_THREADS / 64 - 1) / (FILL_THREADS / 64);
                const uint32_t kend = min(P, (wv + 1) * per);
                for (uint32_t kb = wv * per; kb < kend; kb += 64) {
                    const uint32_t k = kb + lane;
                    uint32_t s0v = 0, s1v = 0, gbv = 0;
                    if (k < kend) { s0v = start[k * NC]; s1v = start[(k + 1) * NC]; gbv = gbase[k]; }
                    const uint32_t np = min(64u, kend - kb);
                    // two runs at a time, one per half of the wave (a run of a fine level holds ~32 items)
#pragma unroll 4
                    for (uint32_t q = 0; q < np; q += 2) {
                        const int src = (int)(q + (lane >> 5));                  // lanes past np hold zeros: an empty run
                        const uint32_t s0 = __shfl(s0v, src, 64), s1 = __shfl(s1v, src, 64), gb = __shfl(gbv, src, 64);
                        for (uint32_t i = s0 + (lane & 31u); i < s1; i += 32) { qvals[gb + (i - s0)] = s_vals[i]; qkeys[gb + (i - s0)] = s_keys[i]; }
                    }
                }
            }
            GRID_STAMP(blockIdx.x, 6);
            GRID_NOTE(blockIdx.x, 7, total);
            __syncthreads();                                   // staging / counters are reused by the next segment
        }
    }
}

// per partition: exclusive scan over the units' counts (in place) and the partition total.  One wavefront per (level, partition).
__global__ __launch_bounds__(256) void k_bwd_scan_units(const int32_t* __restrict__ offsets, uint32_t L, uint32_t U, BwdPlan plan) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= L * BK_MAX) return;
    const uint32_t level = t / BK_MAX, k = t % BK_MAX;
    const uint32_t size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const uint32_t P = (size + PART - 1) >> PART_SHIFT;
    const int lane = threadIdx.x & 63;
    if (P > BK_MAX || k >= P) { if (lane == 0) plan.totals[t] = 0; return; }
    uint32_t* p = plan.part_cnt + (size_t)t * U;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < U; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t v = c < U ? p[c] : 0u;
        const uint32_t inc = lae::wave_incl_scan(v);
        if (c < U) p[c] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) plan.totals[t] = carry;
}

// partition totals -> first queue slot of every partition.  One block.
__global__ __launch_bounds__(1024) void k_bwd_scan_parts(uint32_t n, BwdPlan plan) {
    __shared__ uint32_t lds[17];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? plan.totals[i] : 0;
        uint32_t total;
        const uint32_t ex = lae::block_excl_scan<16>(v, &total, lds);
        if (i < n) plan.offs[i] = carry + ex;
        carry += total;
    }
}

// fp16 bits -> value * 2^24 as a signed integer (exact: finite fp16 values are multiples of 2^-24, |v| * 2^24 < 2^40)
__device__ __forceinline__ long long half_to_fix24(uint32_t u) {
    // branch-free, sign applied to the 11-bit mantissa in 32 bits: (+-mantissa) << (max(e, 1) - 1), subnormals have no hidden bit
    const uint32_t e = (u >> 10) & 31u;
    const int32_t smask = (int32_t)(u << 16) >> 31;
    const uint32_t mant = (u & 1023u) | (min(e, 1u) << 10);
    const int32_t sm = (int32_t)(mant ^ (uint32_t)smask) - smask;
    return (long long)sm << (max(e, 1u) - 1u);
}
// the same for |value| < 128 (exponent field < 22): value * 2^24 fits int32, and the float pipeline converts it exactly
// (fp16 -> fp32 is exact incl. subnormals, the scaling is a power of two, the result an integer below 2^31): 4 instructions
// instead of 10.  Both halves of a word at once.
__device__ __forceinline__ void half2_to_fix24_small(uint32_t w, long long& lo, long long& hi) {
    const half2_t h = __builtin_bit_cast(half2_t, w);
    lo = (long long)(int32_t)((float)h[0] * 16777216.0f);
    hi = (long long)(int32_t)((float)h[1] * 16777216.0f);
}
// value * 2^24 (signed integer) -> nearest fp16 (ties to even), ONE rounding; overflow -> infinity
__device__ __forceinline__ uint32_t fix24_to_half(long long t) {
    const uint32_t sign = t < 0 ? 0x8000u : 0u;
    const unsigned long long mag = t < 0 ? (unsigned long long)(-t) : (unsigned long long)t;
    if (mag < 1024ull) return sign | (uint32_t)mag;              // zero / subnormal: exact
    const int msb = 63 - __builtin_clzll(mag);                   // >= 10
    const int shift = msb - 10;
    unsigned long long q = mag >> shift;                         // 11 bits, leading one included
    if (shift > 0) {
        const unsigned long long rem = mag & ((1ull << shift) - 1ull), halfway = 1ull << (shift - 1);
        if (rem > halfway || (rem == halfway && (q & 1ull))) q++;
    }
    int e = msb - 9;                                             // biased exponent
    if (q == 2048ull) { q = 1024ull; e++; }
    if (e >= 31) return sign | 0x7c00u;
    return sign | ((uint32_t)e << 10) | ((uint32_t)q & 1023u);
}
__device__ __forceinline__ bool half_nonfinite(uint32_t u) { return (u & 0x7c00u) == 0x7c00u; }

template <typename T>
__global__ __launch_bounds__(ACC_THREADS, 8) void k_bwd_acc(
    const int32_t* __restrict__ offsets, T* __restrict__ grad_grid, uint32_t L, LevelScales sc, uint32_t gridtype,
    bool align_corners, BwdPlan plan, const typename BVal<T>::type* __restrict__ qvals, const uint16_t* __restrict__ qkeys,
    unsigned long long* __restrict__ partials, int32_t* __restrict__ nf_flag, uint32_t bk_target) {
    using V = typename BVal<T>::type;
    constexpr bool HALF = sizeof(T) == 2;
    constexpr uint32_t ACCW = HALF ? 2 * PART : PART;
    // fp16 grads: int64 acc0[PART] | acc1[PART] (channel 0 / 1, SoA: fewer bank conflicts than interleaved); fp32: float[2 * PART]
    __shared__ unsigned long long acc64[ACCW];
    __shared__ uint32_t poison[PART / 32];
    __shared__ uint32_t s_cnt[MAX_LEVELS], s_sub[MAX_LEVELS], s_first[MAX_LEVELS];
    __shared__ uint32_t s_ticket, s_arr;
    const uint32_t tid = threadIdx.x;
    if (tid < L) {
        const LevelBins lb = level_bins_of((uint32_t)(offsets[tid + 1] - offsets[tid]), bk_target);
        s_cnt[tid] = lb.P <= BK_MAX ? lb.P * lb.SUB : 0u;
        s_sub[tid] = lb.SUB;
    }
    __syncthreads();
    if (tid == 0) { uint32_t run = 0; for (uint32_t l = 0; l < L; l++) { s_first[l] = run; run += s_cnt[l]; } s_arr = run; }
    __syncthreads();
    const uint32_t total_buckets = s_arr;
    __syncthreads();

    auto apply = [&](uint32_t key, const V& v) {
        const uint32_t e0 = key & (PART - 1), code = key >> 12;
        const uint32_t e1 = e0 ^ ((2u << code) - 1u);
        if constexpr (HALF) {
            const uint32_t w0 = v.x, w1 = v.y;
            // one test for the item's four halves: adding 10 to an exponent field carries out of it iff the field is >= 22,
            // i.e. |value| >= 128 or non-finite; below that the cheap conversion is exact (gradients that large are rare)
            const uint32_t carry = (((w0 & 0x7c007c00u) + 0x28002800u) | ((w1 & 0x7c007c00u) + 0x28002800u)) & 0x80008000u;
            if (carry == 0) {
                // branch-free: a single-corner item (1 in ~4096) carries zeros in its second word, which are added to e0 again
                long long a, b;
                half2_to_fix24_small(w0, a, b);
                atomicAdd(&acc64[e0], (unsigned long long)a);
                atomicAdd(&acc64[PART + e0], (unsigned long long)b);
                const uint32_t e1s = code != KEY_SINGLE ? e1 : e0;
                half2_to_fix24_small(w1, a, b);
                atomicAdd(&acc64[e1s], (unsigned long long)a);
                atomicAdd(&acc64[PART + e1s], (unsigned long long)b);
            } else {
                if (half_nonfinite(w0) || half_nonfinite(w0 >> 16)) atomicOr(&poison[e0 >> 5], 1u << (e0 & 31));
                else {
                    atomicAdd(&acc64[e0], (unsigned long long)half_to_fix24(w0 & 0xffffu));
                    atomicAdd(&acc64[PART + e0], (unsigned long long)half_to_fix24(w0 >> 16));
                }
                if (code != KEY_SINGLE) {
                    if (half_nonfinite(w1) || half_nonfinite(w1 >> 16)) atomicOr(&poison[e1 >> 5], 1u << (e1 & 31));
                    else {
                        atomicAdd(&acc64[e1], (unsigned long long)half_to_fix24(w1 & 0xffffu));
                        atomicAdd(&acc64[PART + e1], (unsigned long long)half_to_fix24(w1 >> 16));
                    }
                }
            }
        } else {
            float* af = reinterpret_cast<float*>(acc64);
            atomicAdd(af + e0, v.x); atomicAdd(af + PART + e0, v.y);
            if (code != KEY_SINGLE) { atomicAdd(af + e1, v.z); atomicAdd(af + PART + e1, v.w); }
        }
    };

    // the first task of every workgroup is its own index (gridDim.x same-address atomics at launch would queue up behind one
    // another); later tasks are drawn from the ticket counter, which therefore counts from gridDim.x
    bool first_task = true;
    [[maybe_unused]] uint32_t stamp_task = 0;
    GRID_STAMP(4096 + blockIdx.x, 0);
    for (;;) {
        if (tid == 0) s_ticket = first_task ? blockIdx.x : atomicAdd(&plan.tickets[0], 1u) + gridDim.x;
        first_task = false;
        __syncthreads();
        const uint32_t t = __builtin_amdgcn_readfirstlane(s_ticket);       // scalar: everything derived from it is wave-uniform
        GRID_STAMP(4096 + blockIdx.x, 1 + stamp_task * 6);                   // ticket in hand
        if (t >= total_buckets) break;
        GRID_NOTE(4096 + blockIdx.x, 2 + stamp_task * 6, t);
        const uint32_t item = t;                                // coarse levels first: their sub-ranges + merge are the longest chains
        uint32_t level = 0;
        while (level + 1 < L && item >= s_first[level + 1]) level++;
        level = __builtin_amdgcn_readfirstlane(level);
        const uint32_t bk = item - __builtin_amdgcn_readfirstlane(s_first[level]);
        const uint32_t SUB = __builtin_amdgcn_readfirstlane(s_sub[level]), p = bk / SUB, sub = bk % SUB;
        const uint32_t table_off = (uint32_t)offsets[level], hashmap_size = (uint32_t)offsets[level + 1] - table_off;
        const uint32_t n = __builtin_amdgcn_readfirstlane(plan.totals[level * BK_MAX + p]), q0 = __builtin_amdgcn_readfirstlane(plan.offs[level * BK_MAX + p]);
        const uint32_t lo = (uint32_t)(((uint64_t)n * sub) / SUB), hi = (uint32_t)(((uint64_t)n * (sub + 1)) / SUB);
        if (n == 0) { __syncthreads(); continue; }              // uniform per (level, partition): no sub-range has work
        // the old values of the lane's table entries: requested now, consumed after the items (their latency hides behind
        // the partition's whole accumulate phase)
        const uint32_t part_lo = p << PART_SHIFT;
        const uint32_t n_ent = min(PART, hashmap_size - part_lo);
        T* __restrict__ dst = grad_grid + ((size_t)table_off + part_lo) * 2;
        uint32_t oldv[HALF ? PART / ACC_THREADS : 1];
        if constexpr (HALF) {
#pragma unroll
            for (int it = 0; it < (int)(PART / ACC_THREADS); it++) { const uint32_t e = tid + it * ACC_THREADS; oldv[it] = e < n_ent ? reinterpret_cast<const uint32_t*>(dst)[e] : 0u; }
        }
        for (uint32_t i = tid; i < ACCW; i += ACC_THREADS) acc64[i] = 0ull;
        if (tid < PART / 32) poison[tid] = 0u;
        __syncthreads();
        GRID_STAMP(4096 + blockIdx.x, 3 + stamp_task * 6);                   // zeroed
        const uint32_t scale_res = (uint32_t)ceilf(sc.scale[level]) + 1;
        const V* __restrict__ pv = qvals + q0;
        const uint16_t* __restrict__ pk = qkeys + q0;
        if (!HALF && scale_res >= COARSE_RES) {
            uint32_t i = lo + tid;
            for (; i + 3 * ACC_THREADS < hi; i += 4 * ACC_THREADS) {              // four loads in flight per lane
                V v[4]; uint32_t k[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { v[j] = pv[i + j * ACC_THREADS]; k[j] = pk[i + j * ACC_THREADS]; }
#pragma unroll
                for (int j = 0; j < 4; j++) apply(k[j], v[j]);
            }
            for (; i < hi; i += ACC_THREADS) { const V v = pv[i]; const uint32_t k = pk[i]; apply(k, v); }
        } else {
            // coarse level: a cell holds many samples of a ray and many rays, so consecutive queue items repeat the same
            // key, and same-address LDS atomics of ONE instruction serialise (level 0: 16 workgroups x 70 k atomics at ~2
            // cycles each = the whole pass).  Each lane takes 8 CONSECUTIVE items and sums them in registers while the key
            // repeats: one atomic group per run, neighbouring lanes 8 items apart.
            if constexpr (HALF) {
                // groups of 8 items on ABSOLUTE 8-item boundaries of the queue: a lane reads one 64-byte line of values and
                // 16 bytes of keys with five 16-byte buffer loads (a partition of a fine level is one round of loads);
                // items of the neighbouring sub-range in the first / last group are masked.  On the coarse levels
                // neighbouring LANES take groups 128 items apart (another ray, usually another cell): lanes 8 items apart sit
                // in the same cell and their atomics to the same 8 entries serialise.
                using u4 = __attribute__((__vector_size__(16))) uint32_t;
                const uint32_t abs_lo = q0 + lo, abs_hi = q0 + hi, A0 = abs_lo & ~7u;
                const uint32_t n_groups = (abs_hi - A0 + 7u) >> 3;
                // whole groups are always inside the queue (its capacity is a multiple of 8 items); the range check of a
                // 16-byte buffer load is all-or-nothing, so the records cover whole groups
                const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<V*>(qvals + A0), 0, (int)(n_groups * 64u), 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkeys + A0), 0, (int)(n_groups * 16u), 0x00020000);
                const bool coarse = scale_res < COARSE_RES;
                for (uint32_t gq = coarse ? (tid & 63u) * (ACC_THREADS / 64) + (tid >> 6) : tid; gq < n_groups; gq += ACC_THREADS) {
                    const u4 va = __builtin_amdgcn_raw_buffer_load_b128(rs_v, (int)(gq * 64u), 0, 0);
                    const u4 vb = __builtin_amdgcn_raw_buffer_load_b128(rs_v, (int)(gq * 64u + 16u), 0, 0);
                    const u4 vc = __builtin_amdgcn_raw_buffer_load_b128(rs_v, (int)(gq * 64u + 32u), 0, 0);
                    const u4 vd = __builtin_amdgcn_raw_buffer_load_b128(rs_v, (int)(gq * 64u + 48u), 0, 0);
                    const u4 kk = __builtin_amdgcn_raw_buffer_load_b128(rs_k, (int)(gq * 16u), 0, 0);
                    const uint32_t w[16] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3], vc[0], vc[1], vc[2], vc[3], vd[0], vd[1], vd[2], vd[3]};
                    const uint32_t a0 = A0 + gq * 8u;
                    const uint32_t first = a0 < abs_lo ? abs_lo - a0 : 0u, last = min(8u, abs_hi - a0);
                    if (!coarse) {             // finer levels: neighbouring items are unrelated, one atomic group per item
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if ((uint32_t)j >= first && (uint32_t)j < last) { V v; v.x = w[2 * j]; v.y = w[2 * j + 1]; apply((kk[j >> 1] >> (16 * (j & 1))) & 0xffffu, v); }
                        continue;
                    }
                    uint32_t ckey = 0;
                    long long s00 = 0, s01 = 0, s10 = 0, s11 = 0;
                    bool bad0 = false, bad1 = false, started = false;
                    auto flush = [&]() {
                        const uint32_t e0 = ckey & (PART - 1), code = ckey >> 12;
                        const uint32_t e1 = e0 ^ ((2u << code) - 1u);
                        if (bad0) atomicOr(&poison[e0 >> 5], 1u << (e0 & 31));
                        else { atomicAdd(&acc64[e0], (unsigned long long)s00); atomicAdd(&acc64[PART + e0], (unsigned long long)s01); }
                        if (code != KEY_SINGLE) {
                            if (bad1) atomicOr(&poison[e1 >> 5], 1u << (e1 & 31));
                            else { atomicAdd(&acc64[e1], (unsigned long long)s10); atomicAdd(&acc64[PART + e1], (unsigned long long)s11); }
                        }
                    };
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if ((uint32_t)j >= first && (uint32_t)j < last) {
                            const uint32_t kj = (kk[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                            if (!started || kj != ckey) {
                                if (started) flush();
                                ckey = kj; s00 = s01 = s10 = s11 = 0; bad0 = bad1 = false; started = true;
                            }
                            const uint32_t w0 = w[2 * j], w1 = w[2 * j + 1];
                            const uint32_t carry = (((w0 & 0x7c007c00u) + 0x28002800u) | ((w1 & 0x7c007c00u) + 0x28002800u)) & 0x80008000u;
                            if (carry == 0) {            // every half below 128 in magnitude: exact through the float pipeline
                                long long a, b;
                                half2_to_fix24_small(w0, a, b); s00 += a; s01 += b;
                                half2_to_fix24_small(w1, a, b); s10 += a; s11 += b;
                            } else {
                                if (half_nonfinite(w0) || half_nonfinite(w0 >> 16)) bad0 = true;
                                else { s00 += half_to_fix24(w0 & 0xffffu); s01 += half_to_fix24(w0 >> 16); }
                                if (half_nonfinite(w1) || half_nonfinite(w1 >> 16)) bad1 = true;
                                else { s10 += half_to_fix24(w1 & 0xffffu); s11 += half_to_fix24(w1 >> 16); }
                            }
                        }
                    }
                    if (started) flush();
                }
            } else {
                for (uint32_t c = lo + tid * 8u; c < hi; c += ACC_THREADS * 8u) {
                    V v[8]; uint32_t k[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { const uint32_t ii = min(c + (uint32_t)j, hi - 1u); v[j] = pv[ii]; k[j] = pk[ii]; }
                    const uint32_t cnt = min(8u, hi - c);
#pragma unroll
                    for (int j = 0; j < 8; j++) if ((uint32_t)j < cnt) apply(k[j], v[j]);
                }
            }
        }
        GRID_STAMP(4096 + blockIdx.x, 4 + stamp_task * 6);                   // thread 0 out of the items
        __syncthreads();
        GRID_STAMP(4096 + blockIdx.x, 5 + stamp_task * 6);                   // every wave out of the items
        if (SUB > 1) {
            // ---- sub-ranges of one partition: every one ADDS its exact partial sums (int64: the adds commute, so the
            // result has the same bits whatever the order) into the partition's merge record with device-scope atomics;
            // the last to arrive (told by the value its counter add returned) takes the sums out with atomic exchanges,
            // which also leaves the record zero for the next launch.  Every access to the record is an atomic, so no
            // fence and no cache maintenance is needed (an agent-scope release here writes back ALL dirty lines of the
            // XCD's L2 -- tens of MB of queue right after the fill pass: 40 us per level measured; storing 64 KB partials
            // write-through and letting the last arriver add SUB of them: 16 us on its critical path).
            unsigned long long* __restrict__ rec = partials + (size_t)(level * SUB_RECS + p) * sub_rec_words<T>();
            uint32_t* __restrict__ rec_poison = reinterpret_cast<uint32_t*>(rec + ACCW);
            if constexpr (HALF) {
                for (uint32_t i = tid; i < ACCW; i += ACC_THREADS) { const unsigned long long v = acc64[i]; if (v) atomicAdd(rec + i, v); }
            } else {
                const float* af = reinterpret_cast<const float*>(acc64);
                float* rf = reinterpret_cast<float*>(rec);
                for (uint32_t i = tid; i < 2 * PART; i += ACC_THREADS) { const float v = af[i]; if (v != 0.0f) atomicAdd(rf + i, v); }
            }
            if (tid < PART / 32 && poison[tid]) atomicOr(rec_poison + tid, poison[tid]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) s_arr = atomicAdd(&plan.tickets[TICKET_ARRIVALS + level * SUB_RECS + p], 1u);
            __syncthreads();
            if (s_arr != SUB - 1) { GRID_STAMP(4096 + blockIdx.x, 6 + stamp_task * 6); stamp_task++; __syncthreads(); continue; }
            for (uint32_t i = tid; i < ACCW; i += ACC_THREADS) acc64[i] = atomicExch(rec + i, 0ull);
            if (tid < PART / 32) poison[tid] = atomicExch(rec_poison + tid, 0u);
            __syncthreads();
        }
        if constexpr (HALF) {
            uint32_t* d2 = reinterpret_cast<uint32_t*>(dst);
            bool wrote_nonfinite = false;
            // the old values of the lane's entries first, all in flight together (a load per touched entry inside the loop
            // below made every round wait for its own memory latency)
#pragma unroll
            for (int it = 0; it < (int)(PART / ACC_THREADS); it++) {
                const uint32_t e = tid + it * ACC_THREADS;
                if (e >= n_ent) break;
                const long long i0 = (long long)acc64[e], i1 = (long long)acc64[PART + e];
                const bool bad = (poison[e >> 5] >> (e & 31)) & 1u;
                if (i0 == 0 && i1 == 0 && !bad) continue;
                const uint32_t o = oldv[it];               // only writer of this table slice: old + exact sum, rounded ONCE
                uint32_t r0, r1;
                if (bad) { r0 = 0x7e00u; r1 = 0x7e00u; }
                else {
                    r0 = half_nonfinite(o) ? (o & 0xffffu) : fix24_to_half(i0 + half_to_fix24(o & 0xffffu));
                    r1 = half_nonfinite(o >> 16) ? (o >> 16) : fix24_to_half(i1 + half_to_fix24(o >> 16));
                }
                d2[e] = r0 | (r1 << 16);
                wrote_nonfinite |= half_nonfinite(r0) || half_nonfinite(r1);
            }
            // every non-finite value in the table gradient was stored by a flush like this one: the caller's flag (the
            // optimizer's found_inf word) makes a scan of the 12 M-entry gradient for non-finite values unnecessary
            if (nf_flag && wrote_nonfinite) atomicOr(nf_flag, 1);
        } else {
            const float* af = reinterpret_cast<const float*>(acc64);
            float2* d2 = reinterpret_cast<float2*>(dst);
            for (uint32_t e = tid; e < n_ent; e += ACC_THREADS) {
                const float v0 = af[e], v1 = af[PART + e];
                if (v0 == 0.0f && v1 == 0.0f) continue;
                float2 o = d2[e]; o.x += v0; o.y += v1; d2[e] = o;
                if (nf_flag && !(isfinite(o.x) && isfinite(o.y))) atomicOr(nf_flag, 1);
            }
        }
        GRID_STAMP(4096 + blockIdx.x, 6 + stamp_task * 6);                   // flushed
        stamp_task++;
        __syncthreads();
    }
}

// [L][B][2] -> [B][L][2] through an LDS tile: the gather kernel writes level-major (each level's block stores 256
// consecutive pairs = full lines); storing 4 B per lane at a 64 B stride straight into [B, L*2] measured 123 MB of
// HBM writes for 16 MB of output (profiles/r1_pmc_fetch_write_per_kernel.csv).
template <typename T>
__global__ __launch_bounds__(256) void k_out_transpose(const T* __restrict__ in, T* __restrict__ out, uint32_t B, uint32_t L) {
    using V = typename std::conditional<sizeof(T) == 2, uint32_t, uint2>::type;
    __shared__ V tile[256 * 33];
    const uint32_t b0 = blockIdx.x * 256;
    const uint32_t nb = min(256u, B - b0);
    const V* src = reinterpret_cast<const V*>(in);
    for (uint32_t e = threadIdx.x; e < nb * L; e += 256) {
        const uint32_t l = e / nb, b = e % nb;
        tile[b * (L + 1) + l] = src[(size_t)l * B + b0 + b];
    }
    __syncthreads();
    V* dstv = reinterpret_cast<V*>(out) + (size_t)b0 * L;
    for (uint32_t e = threadIdx.x; e < nb * L; e += 256) dstv[e] = tile[(e / L) * (L + 1) + (e % L)];
}

// [B][L][2] -> [L][B][2] through an LDS tile (256 samples x L levels), coalesced on both sides
template <typename T>
__global__ __launch_bounds__(256) void k_grad_transpose(const T* __restrict__ in, T* __restrict__ out, uint32_t B, uint32_t L) {
    using V = typename std::conditional<sizeof(T) == 2, uint32_t, uint2>::type;     // one (c0, c1) pair
    __shared__ V tile[256 * 33];
    const uint32_t b0 = blockIdx.x * 256;
    const uint32_t nb = min(256u, B - b0);
    const V* src = reinterpret_cast<const V*>(in) + (size_t)b0 * L;
    for (uint32_t e = threadIdx.x; e < nb * L; e += 256) tile[(e / L) * (L + 1) + (e % L)] = src[e];
    __syncthreads();
    V* dstv = reinterpret_cast<V*>(out);
    for (uint32_t e = threadIdx.x; e < nb * L; e += 256) {
        const uint32_t l = e / nb, b = e % nb;
        dstv[(size_t)l * B + b0 + b] = tile[b * (L + 1) + l];
    }
}

// ---------------------------------------------------------------- K15
// gridencoder.cu:343-369
template <typename T>
__global__ void k_grid_input_bwd(const T* __restrict__ grad, const T* __restrict__ dy_dx, T* __restrict__ grad_inputs,
                                 uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint64_t gs_b, uint64_t gs_l) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* dd = dy_dx + (size_t)b * L * D * C;
    T r = (T)0.0f;
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t ch = 0; ch < C; ch++) {
            const T gv = grad[(size_t)b * gs_b + (size_t)l * gs_l + ch];
            const T dv = dd[(size_t)l * D * C + d * C + ch];
            if constexpr (sizeof(T) == 2) r = (half_t)((float)r + (float)(half_t)((float)gv * (float)dv));
            else r = fmaf(gv, dv, r);
        }
    grad_inputs[t] = r;
}

// ---------------------------------------------------------------- K16
// gridencoder.cu:506-610 (fp32 only: grid.py:165 runs it with autocast disabled)
template <int D, int C>
__global__ __launch_bounds__(GRID_BLOCK) void k_grad_tv(const float* __restrict__ inputs, const float* __restrict__ grid,
                                                         float* __restrict__ grad, const int32_t* __restrict__ offsets,
                                                         float weight, uint32_t B, uint32_t L, LevelScales sc,
                                                         uint32_t gridtype, bool align_corners, uint32_t nb) {
    uint32_t level, chunk;
    block_to_level_chunk(nb, false, level, chunk);
    if (level >= L) return;
    const uint32_t b = chunk * GRID_BLOCK + threadIdx.x;
    if (b >= B) return;
    const LevelInfo<D> li = level_info<D>(sc, offsets, level, gridtype, align_corners);
    const float* tab = grid + (size_t)li.table_off * C;
    float* gtab = grad + (size_t)li.table_off * C;
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float xv = inputs[(size_t)b * D + d];
        if (xv < 0.0f || xv > 1.0f) return;
        pg[d] = (uint32_t)floorf(fmaf(xv, li.scale, align_corners ? 0.0f : 0.5f));
    }
    float res[C], idelta[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) { res[ch] = 0; idelta[ch] = 0; }
    const uint32_t index = cell_index<D>(li, pg) * C;
    const float w = weight / (float)(2 * D);
#pragma unroll
    for (int d = 0; d < D; d++) {
        const uint32_t cur = pg[d];
        if (cur < li.resolution) {
            pg[d] = cur + 1;
            const uint32_t ir = cell_index<D>(li, pg) * C;
#pragma unroll
            for (int ch = 0; ch < C; ch++) { const float gv = tab[index + ch] - tab[ir + ch]; res[ch] += gv; idelta[ch] += gv * gv; }
        }
        if (cur > 0) {
            pg[d] = cur - 1;
            const uint32_t il = cell_index<D>(li, pg) * C;
#pragma unroll
            for (int ch = 0; ch < C; ch++) { const float gv = tab[index + ch] - tab[il + ch]; res[ch] += gv; idelta[ch] += gv * gv; }
        }
        pg[d] = cur;
    }
#pragma unroll
    for (int ch = 0; ch < C; ch++) atomicAdd(gtab + index + ch, w * res[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f)));
}

// ---------------------------------------------------------------- host dispatch
static int fill_scales(LevelScales& sc, uint32_t L, float S, uint32_t H) {
    if (L == 0 || L > MAX_LEVELS) return LAE_EINVAL;
    for (uint32_t l = 0; l < L; l++) sc.scale[l] = fmaf(exp2f((float)l * S), (float)H, -1.0f);   // :138
    for (uint32_t l = L; l < MAX_LEVELS; l++) sc.scale[l] = 0.f;
    sc.in_shift = 0.0f; sc.in_scale = 1.0f;
    return LAE_OK;
}

struct FwdArgs {
    const float* inputs; const void* emb; const int32_t* offsets; void* out; uint32_t B, L; LevelScales sc;
    void* dy_dx; uint32_t gridtype; bool align; uint32_t interp; uint64_t os_b, os_l; hipStream_t stream;
    const uint32_t* B_dev = nullptr; uint32_t B_launch = 0;     // frame loop: device-side row count, host bound for the launch
    uint32_t B_likely = 0;                                      // frame loop: the rows the host expects (<= B_launch): the scheduled launch covers them, k_grid_fwd_lean_tail the rest
    const int32_t* offsets_host = nullptr;                      // the caller's host copy of `offsets` (L + 1 ints) or NULL
    const float* level_cost = nullptr;                          // frame loop: measured per-level chunk costs (balance only) or NULL
};

// ---- schedule of k_grid_fwd_lean (see the kernel's header).  Relative cost of one chunk of a level, calibrated on the
// bench batches (tools/ubench/grid_fwd_variants.hip, then a sweep with tools/grid_fwd_bench.py on one-view and 16-view
// batches): dense 1; hashed 1.25 up to resolution ~80, rising with log2(resolution) to 2.3 at ~550 (consecutive samples of a
// ray stop sharing cache lines) and 4.5 from ~1000 on (every corner row is its own L2 request).  Only the balance depends on it, never a result.
// (round 3, tools/grid_fwd_spans.py -- per-block stamps of one launch: with this table the XCDs end at 38.5-41 us (one finest level
// each) and 45-47 us (two middle levels + dense eighths); a table re-fitted to those spans (middle levels 2.0-2.9 instead of
// 1.4-2.4) moves the late end to the XCDs that carry levels 11 / 12 + dense eighths and the launch still spans 46.5 us: two levels
// sharing an XCD overlap better than their solo costs add up, and the dense eighths run last on every XCD.  Kept as calibrated.)
static float fwd_level_cost(bool hashed, uint32_t resolution) {
    if (!hashed) return 1.0f;
    const float lr = log2f((float)resolution);
    if (lr <= 6.3f) return 1.25f;
    if (lr <= 9.1f) return 1.25f + (lr - 6.3f) * (1.05f / 2.8f);
    if (lr <= 10.0f) return 2.3f + (lr - 9.1f) * (2.2f / 0.9f);
    return 4.5f;
}
// The same for a FRAME's rows (ray-major: neighbouring rows are neighbouring samples of one ray, neighbouring rays neighbouring
// pixels), from per-block stamps of frame-loop launches (tools/frame_grid_spans.py, profiles/r4_frame_grid_spans.txt): a level
// alone on an XCD took 27 / 32 / 33 / 39 / 40 / 44 / 46 / 54 / 58 / 71 / 83 us for resolutions 80 ... 2048 and a dense level 21 --
// the finest levels cost 2.8-4x a dense one, not 4.5x, and rise steadily instead of jumping at ~1000: with the training table
// the XCDs that carry ONE fine level finished at 58 / 71 us of an 87 us launch.
static float fwd_level_cost_frame(bool hashed, uint32_t resolution) {
    if (!hashed) return 1.0f;
    const float lr = log2f((float)resolution);
    if (lr <= 6.3f) return 1.29f;
    if (lr <= 9.1f) return 1.29f + (lr - 6.3f) * (0.91f / 2.8f);
    if (lr <= 11.0f) return 2.2f + (lr - 9.1f) * (1.75f / 1.9f);
    static const float top_slope = [] { const char* e = getenv("LAE_GRID_FWD_FRAME_SLOPE"); return e ? (float)atof(e) : 0.5f; }();   // beyond resolution 2048 (bound-2 scenes: finest levels 2819 / 4096) the rise flattens: slope 0.92 / 0.5 / 0.2 / 0 -> 1080p frame 60.8 / 59.7 / 59.9 / 61.4 ms
    return 3.95f + (lr - 11.0f) * top_slope;
}
static int g_fwd_frame_sched = -1;                         // LAE_GRID_FWD_FRAME_SCHED: 0 training table, 1 (default) frame table, 2 frame table + the costliest levels in halves
// (measured, 800x800 / 1080p / one rank's shard of it: 0: 10.9 / 67.2 / 11.15 ms, 1: 10.95 / 62.2 / 10.9, 2: 11.1 / 63.7 / 10.9 -- a level in two
// halves costs 2 x 48 us against 83 whole: its table then streams into two L2s)
static void fwd_sched_default(FwdSched& fs, uint32_t L, uint32_t nb) {     // level l on XCD l mod 8, one level at a time
    for (int x = 0; x < 8; x++) fs.nseg[x] = 0;
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t x = l & 7u;
        fs.seg[x][fs.nseg[x]++] = FwdSeg{l, 0u, nb, 0u};
    }
}
// returns the largest number of blocks any XCD owns
static uint32_t fwd_sched_build(FwdSched& fs, uint32_t L, uint32_t nb, const LevelScales& sc, const int32_t* offs, bool frame = false,
                                const float* cost_override = nullptr) {
    bool ok = offs != nullptr && L <= 32 && nb >= 64;
    float cost[MAX_LEVELS]; bool dense[MAX_LEVELS];
    if (g_fwd_frame_sched < 0) { const char* e = getenv("LAE_GRID_FWD_FRAME_SCHED"); g_fwd_frame_sched = e ? atoi(e) : 1; }
    const int fmode = frame ? g_fwd_frame_sched : 0;
    if (ok) {
        float total = 0.f;
        for (uint32_t l = 0; l < L; l++) {
            const uint32_t res = (uint32_t)ceilf(sc.scale[l]) + 1;
            const uint64_t size = (uint64_t)(offs[l + 1] - offs[l]);
            const uint64_t full = (uint64_t)(res + 1) * (res + 1) * (res + 1);
            dense[l] = full <= size;
            cost[l] = fmode ? fwd_level_cost_frame(!dense[l], res) : fwd_level_cost(!dense[l], res);
            if (frame && cost_override) cost[l] = cost_override[l];
            total += cost[l];
        }
        // a whole hashed level is one item (its 2 MB table then lives in ONE L2); in a frame a level that alone exceeds an XCD's
        // fair share would set the launch's span, so it goes out as two halves of its chunk range (two L2s hold its table)
        struct Item { float cost; uint32_t level; bool piece; uint32_t c0, n; };
        std::vector<Item> items;
        const uint32_t piece = lae::cdiv(nb, 8u);
        for (uint32_t l = 0; l < L; l++) {
            if (!dense[l]) {
                if (fmode == 2 && cost[l] > 0.97f * total / 8.0f) {
                    const uint32_t h = nb / 2;
                    items.push_back(Item{cost[l] * h, l, false, 0u, h});
                    items.push_back(Item{cost[l] * (nb - h), l, false, h, nb - h});
                } else items.push_back(Item{cost[l] * nb, l, false, 0u, nb});
            }
            else for (uint32_t k = 0; k < 8 && k * piece < nb; k++) items.push_back(Item{cost[l] * std::min(piece, nb - k * piece), l, true, 0u, 0u});
        }
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.cost > b.cost; });
        float load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t pieces[8][MAX_LEVELS] = {};
        for (int x = 0; x < 8; x++) fs.nseg[x] = 0;
        for (const Item& it : items) {
            int best = 0;
            for (int x = 1; x < 8; x++) if (load[x] < load[best]) best = x;
            load[best] += it.cost;
            if (it.piece) pieces[best][it.level]++;
            else if (fs.nseg[best] < FWD_MAX_SEG) fs.seg[best][fs.nseg[best]++] = FwdSeg{it.level, it.c0, it.n, 0u};
            else ok = false;
        }
        for (uint32_t l = 0; l < L && ok; l++) {
            if (!dense[l]) continue;
            uint32_t c0 = 0;
            if (fmode) {
                // a frame's launch is sized by a host BOUND of the row count: the chunks beyond the live rows are empty, and with
                // contiguous eighths the XCDs holding the last ones would idle while the first ones do all of a dense level
                // (1080p frame: 86-93 us of dense work on two XCDs, 6 on another) -- an XCD takes every 8th chunk instead
                for (int x = 0; x < 8; x++) {
                    if (!pieces[x][l]) continue;
                    if (fs.nseg[x] < FWD_MAX_SEG) fs.seg[x][fs.nseg[x]++] = FwdSeg{l, c0, lae::cdiv(nb, 8u) * pieces[x][l], pieces[x][l]}; else ok = false;
                    c0 += pieces[x][l];
                }
                if (c0 != std::min(8u, lae::cdiv(nb, piece))) ok = false;
                continue;
            }
            for (int x = 0; x < 8; x++) {
                if (!pieces[x][l]) continue;
                const uint32_t n = std::min(pieces[x][l] * piece, nb - c0);
                if (fs.nseg[x] < FWD_MAX_SEG) fs.seg[x][fs.nseg[x]++] = FwdSeg{l, c0, n, 0u}; else ok = false;
                c0 += n;
            }
            if (c0 != nb) ok = false;
        }
    }
    if (!ok) fwd_sched_default(fs, L, nb);
    uint32_t mx = 0;
    for (int x = 0; x < 8; x++) {
        uint32_t t = 0;
        for (uint32_t q = 0; q < fs.nseg[x]; q++) t += fs.seg[x][q].n;
        mx = std::max(mx, t);
    }
    return mx;
}
static int g_fwd_mode = 0;                                 // 0: lean kernel + balanced schedule, 1: lean + (l, l+8) map, 2: generic kernel

template <typename T, int D, int C>
static void launch_fwd(const FwdArgs& a) {
    const uint32_t nb_safe = lae::cdiv(a.B_dev ? a.B_launch : a.B, GRID_BLOCK);
    // frame, lean kernel: the scheduled launch is sized for the expected rows (a multiple of 8 chunks: strided dense pieces)
    // -- when that spares enough workgroups to pay for the second launch (~4 us in the iteration's chain against ~0.3 ns per
    // workgroup that finds no rows, L of them per chunk: from ~2000 chunks on; an 800x800 frame stays with one launch)
    static const uint32_t tail_min_chunks = [] { const char* e = getenv("LAE_GRID_FWD_TAIL_MIN"); return e ? (uint32_t)atoi(e) : 2048u; }();
    uint32_t nb = a.B_dev && a.B_likely ? std::min((lae::cdiv(a.B_likely, GRID_BLOCK) + 7u) / 8u * 8u, nb_safe) : nb_safe;
    if (nb_safe - nb < tail_min_chunks) nb = nb_safe;
    if constexpr (std::is_same<T, half_t>::value && D == 3 && C == 2) {
        if (g_fwd_mode != 2 && !a.dy_dx && a.interp == 0 && !a.align && a.gridtype == 0 && a.L <= 8 * FWD_MAX_SEG && a.L <= MAX_LEVELS) {
            FwdSched fs;
            static const std::vector<float> env_cost = [] {      // probe: LAE_GRID_FWD_COSTS="c0,c1,...": per-level chunk costs of frame launches
                std::vector<float> v; const char* e = getenv("LAE_GRID_FWD_COSTS");
                while (e && *e) { char* end; v.push_back(strtof(e, &end)); if (end == e) break; e = *end ? end + 1 : end; }
                return v; }();
            const float* co = a.level_cost ? a.level_cost : (env_cost.size() >= a.L ? env_cost.data() : nullptr);
            const uint32_t per_xcd = fwd_sched_build(fs, a.L, nb, a.sc, g_fwd_mode == 0 ? a.offsets_host : nullptr, a.B_dev != nullptr, co);
            k_grid_fwd_lean<<<per_xcd * 8, GRID_BLOCK, 0, a.stream>>>(a.inputs, (const half_t*)a.emb, a.offsets, (half_t*)a.out, std::min(a.B, nb * (uint32_t)GRID_BLOCK), a.sc,
                                                                      fs, a.os_b, a.os_l, a.B_dev
#ifdef LAE_GRID_FWD_LOOP_PROBE
                                                                      , 0u
#endif
                                                                      );
            if (nb < nb_safe)                                  // rows beyond the expected ones, if the device has any
                k_grid_fwd_lean_tail<<<nb_safe - nb, GRID_BLOCK, 0, a.stream>>>(a.inputs, (const half_t*)a.emb, a.offsets, (half_t*)a.out, a.B, a.sc, a.L,
                                                                                 a.os_b, a.os_l, a.B_dev, nb * (uint32_t)GRID_BLOCK);
            return;
        }
    }
    const bool xcd = (a.L % 8) == 0;
    k_grid_fwd<T, D, C><<<nb_safe * a.L, GRID_BLOCK, 0, a.stream>>>(a.inputs, (const T*)a.emb, a.offsets, (T*)a.out, a.B, a.L,
                                                                a.sc, (T*)a.dy_dx, a.gridtype, a.align, a.interp, nb_safe,
                                                                xcd, a.os_b, a.os_l, a.B_dev);
}
template <typename T, int D>
static int dispatch_fwd_c(const FwdArgs& a, uint32_t C) {
    switch (C) {
        case 1: launch_fwd<T, D, 1>(a); return LAE_OK;
        case 2: launch_fwd<T, D, 2>(a); return LAE_OK;
        case 4: launch_fwd<T, D, 4>(a); return LAE_OK;
        case 8: launch_fwd<T, D, 8>(a); return LAE_OK;
        default: return LAE_EINVAL;     // gridencoder.cu:381 "C must be 1, 2, 4, or 8"
    }
}
template <typename T>
static int dispatch_fwd_d(const FwdArgs& a, uint32_t D, uint32_t C) {
    switch (D) {
        case 2: return dispatch_fwd_c<T, 2>(a, C);
        case 3: return dispatch_fwd_c<T, 3>(a, C);
        case 4: return dispatch_fwd_c<T, 4>(a, C);
        case 5: return dispatch_fwd_c<T, 5>(a, C);
        default: return LAE_EINVAL;     // gridencoder.cu:398
    }
}

struct BwdArgs {
    const void* grad; const float* inputs; const int32_t* offsets; void* gemb; uint32_t B, L; LevelScales sc;
    uint32_t gridtype; bool align; uint32_t interp; uint64_t gs_b, gs_l; hipStream_t stream;
    const int32_t* offsets_host = nullptr;
    int32_t* nf_flag = nullptr;                                 // set to 1 when a non-finite table gradient is stored (binned path)
    uint32_t* touched = nullptr;                                // bit per 8 table entries (one 64-byte line of fp32 pairs): set by the count pass
    uint32_t* touched_full = nullptr;                           // MAX_LEVELS words behind the bitmap: level l is fully marked
};
template <typename T, int D, int C>
static void launch_bwd(const BwdArgs& a) {
    const uint32_t nb = lae::cdiv(a.B, GRID_BLOCK);
    const bool xcd = (a.L % 8) == 0;
    k_grid_bwd<T, D, C><<<nb * a.L, GRID_BLOCK, 0, a.stream>>>(a.grad ? (const T*)a.grad : nullptr, a.inputs, a.offsets,
                                                                (T*)a.gemb, a.B, a.L, a.sc, a.gridtype, a.align,
                                                                a.interp, nb, xcd, a.gs_b, a.gs_l, false);
}
template <typename T, int D>
static int dispatch_bwd_c(const BwdArgs& a, uint32_t C) {
    if constexpr (sizeof(T) == 2) {
        switch (C) {     // odd C in fp16 is rejected, grid.py:42-44 never produces it
            case 2: launch_bwd<T, D, 2>(a); return LAE_OK;
            case 4: launch_bwd<T, D, 4>(a); return LAE_OK;
            case 8: launch_bwd<T, D, 8>(a); return LAE_OK;
            default: return LAE_EINVAL;
        }
    } else {
        switch (C) {
            case 1: launch_bwd<T, D, 1>(a); return LAE_OK;
            case 2: launch_bwd<T, D, 2>(a); return LAE_OK;
            case 4: launch_bwd<T, D, 4>(a); return LAE_OK;
            case 8: launch_bwd<T, D, 8>(a); return LAE_OK;
            default: return LAE_EINVAL;
        }
    }
}
template <typename T>
static int dispatch_bwd_d(const BwdArgs& a, uint32_t D, uint32_t C) {
    switch (D) {
        case 2: return dispatch_bwd_c<T, 2>(a, C);
        case 3: return dispatch_bwd_c<T, 3>(a, C);
        case 4: return dispatch_bwd_c<T, 4>(a, C);
        case 5: return dispatch_bwd_c<T, 5>(a, C);
        default: return LAE_EINVAL;
    }
}

static int grid_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                        uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx, uint32_t gridtype,
                        int align_corners, uint32_t interp, int dtype, bool blc, void* stream, float in_shift = 0.0f,
                        float in_scale = 1.0f, const int32_t* offsets_host = nullptr) {
    if (B == 0) return LAE_OK;
    if (!inputs || !embeddings || !offsets || !outputs) return LAE_ENULL;
    if (gridtype > 1 || interp > 1) return LAE_EINVAL;
    FwdArgs a;
    a.offsets_host = offsets_host;
    a.inputs = inputs; a.emb = embeddings; a.offsets = offsets; a.out = outputs; a.B = B; a.L = L;
    int rc = fill_scales(a.sc, L, S, H);
    if (rc) return rc;
    a.sc.in_shift = in_shift; a.sc.in_scale = in_scale;
    a.dy_dx = dy_dx; a.gridtype = gridtype; a.align = align_corners != 0; a.interp = interp;
    a.stream = reinterpret_cast<hipStream_t>(stream);
    if (dtype != LAE_F32 && dtype != LAE_F16) return LAE_EINVAL;
    // [B, L*C] output with C == 2: gather level-major into the workspace, then one tiled transpose
    const bool staged = blc && C == 2 && L <= 32;
    void* lbc = nullptr;
    if (staged) {
        lbc = lae::workspace(lae::WS_GRID_OUT_T, (size_t)B * L * C * (dtype == LAE_F16 ? 2 : 4), a.stream);
        if (!lbc) return LAE_ELAUNCH;
        a.out = lbc;
    }
    const bool level_major = !blc || staged;
    a.os_b = level_major ? C : (uint64_t)L * C;
    a.os_l = level_major ? (uint64_t)B * C : C;
    if (dtype == LAE_F32) rc = dispatch_fwd_d<float>(a, D, C);
    else rc = dispatch_fwd_d<half_t>(a, D, C);
    if (rc) return rc;
    if (staged) {
        if (dtype == LAE_F16) k_out_transpose<half_t><<<lae::cdiv(B, 256), 256, 0, a.stream>>>((const half_t*)lbc, (half_t*)outputs, B, L);
        else k_out_transpose<float><<<lae::cdiv(B, 256), 256, 0, a.stream>>>((const float*)lbc, (float*)outputs, B, L);
    }
    return lae::check_launch("grid_encode_forward");
}

// The binned backward in two halves.  PLAN (count pass + scans) reads the sample positions only and fills {tickets |
// partition totals | partition offsets | per-unit counter rows | per-partition unit table}; EXEC (fill + accumulate) needs the
// gradients.  A caller may run PLAN early -- e.g. right after the march, on another stream beside the forward pass
// (lae_grid_encode_backward_plan) -- the default entry points run both back to back with the plan in the library workspace.
// Library workspace of EXEC: {partial sums of sub-ranges | queue: values, keys}; the queue is sized for the worst case of 8
// items per (sample, level) -- hashed levels produce 4 -- and only its front is ever touched.
struct PlanLayout { size_t totals_off, offs_off, cnt_off, part_off, bytes; };
static inline uint32_t bwd_units(uint32_t B) { return lae::cdiv(B, UNIT_SAMPLES * SEGS) * SEGS; }
static inline PlanLayout plan_layout(uint32_t B, uint32_t L) {
    const size_t U = bwd_units(B);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    PlanLayout w;
    w.totals_off = up((size_t)TICKET_WORDS * 4);
    w.offs_off = w.totals_off + up((size_t)L * BK_MAX * 4);
    w.cnt_off = w.offs_off + up((size_t)L * BK_MAX * 4);
    w.part_off = w.cnt_off + up((size_t)L * U * BK_MAX * 2);
    w.bytes = w.part_off + up((size_t)L * BK_MAX * U * 4);
    return w;
}
static inline BwdPlan plan_at(void* buf, uint32_t B, uint32_t L) {
    const PlanLayout lay = plan_layout(B, L);
    uint8_t* p = reinterpret_cast<uint8_t*>(buf);
    BwdPlan plan;
    plan.tickets = reinterpret_cast<uint32_t*>(p);
    plan.totals = reinterpret_cast<uint32_t*>(p + lay.totals_off);
    plan.offs = reinterpret_cast<uint32_t*>(p + lay.offs_off);
    plan.cnt = reinterpret_cast<uint16_t*>(p + lay.cnt_off);
    plan.part_cnt = reinterpret_cast<uint32_t*>(p + lay.part_off);
    return plan;
}
// merge records of split partitions: at the START of the workspace, sized for MAX_LEVELS whatever the call (their place
// must not move with B or L: they are zeroed once per buffer and every launch leaves them zero)
static inline size_t merge_bytes() { return ((size_t)MAX_LEVELS * SUB_RECS * sub_rec_words<half_t>() * 8 + 255) / 256 * 256; }
template <typename T>
static inline size_t bwd_exec_ws_bytes(uint32_t B, uint32_t L) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t cap = (size_t)L * bwd_units(B) * UNIT_SAMPLES * 8;
    return merge_bytes() + up(cap * sizeof(typename BVal<T>::type)) + up(cap * 2);
}

template <typename T>
static void bwd_plan(const float* inputs, const int32_t* offsets, uint32_t B, uint32_t L, const BwdArgs& a, const BwdPlan& plan) {
    const uint32_t U = bwd_units(B);
    // the count pass keeps the line bits of its level in LDS when the caller wants its "ever touched" bitmap maintained:
    // room for the largest level (a.touched implies a.offsets_host and every level <= 2^21 entries, checked by the callers)
    size_t line_lds = 0;
    if (a.touched) {
        uint32_t words = ((1u << 21) >> 3 >> 5) + 2;
        if (a.offsets_host) {
            words = 1;
            for (uint32_t l = 0; l < L; l++) {
                const uint32_t off = (uint32_t)a.offsets_host[l], size = (uint32_t)(a.offsets_host[l + 1] - a.offsets_host[l]);
                words = std::max(words, ((((off >> 3) & 31u) + ((size + 7u) >> 3) + 31u) >> 5));
            }
        }
        line_lds = (size_t)words * 4;
    }
    if (a.gridtype == 0 && !a.align && a.interp == 0)
        k_bwd_walk<T, false, true><<<U / SEGS * L, FILL_THREADS, line_lds, a.stream>>>(nullptr, inputs, offsets, B, L, a.sc, a.gridtype, a.align, a.interp, U, plan, nullptr, nullptr, a.touched, a.touched_full);
    else
        k_bwd_walk<T, false, false><<<U / SEGS * L, FILL_THREADS, line_lds, a.stream>>>(nullptr, inputs, offsets, B, L, a.sc, a.gridtype, a.align, a.interp, U, plan, nullptr, nullptr, a.touched, a.touched_full);
    k_bwd_scan_units<<<lae::cdiv((size_t)L * BK_MAX, 4), 256, 0, a.stream>>>(offsets, L, U, plan);
    k_bwd_scan_parts<<<1, 1024, 0, a.stream>>>(L * BK_MAX, plan);
}

template <typename T>
static int launch_bwd_fast(const void* gT, const float* inputs, const int32_t* offsets, void* gemb, uint32_t B, uint32_t L,
                           const BwdArgs& a, const void* caller_plan = nullptr) {
    using V = typename BVal<T>::type;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const uint32_t U = bwd_units(B);
    const size_t plan_bytes = caller_plan ? 0 : plan_layout(B, L).bytes;
    uint8_t* ws = reinterpret_cast<uint8_t*>(lae::workspace(lae::WS_GRID_BINS, plan_bytes + bwd_exec_ws_bytes<T>(B, L), a.stream));
    if (!ws) return LAE_ELAUNCH;
    unsigned long long* partials = reinterpret_cast<unsigned long long*>(ws);
    {   // the merge records must start at zero (afterwards every launch leaves them zero): once per buffer
        static std::mutex mtx;
        static std::vector<void*> zeroed;
        static uint64_t epoch = 0;
        std::lock_guard<std::mutex> lk(mtx);
        if (epoch != lae::workspace_epoch()) { zeroed.clear(); epoch = lae::workspace_epoch(); }
        if (std::find(zeroed.begin(), zeroed.end(), (void*)ws) == zeroed.end()) {
            if (hipMemsetAsync(ws, 0, merge_bytes(), a.stream) != hipSuccess) {
                (void)hipGetLastError();
                lae::set_last_error_str("grid backward: could not zero the merge records (first call inside a capture? warm up eagerly)");
                return LAE_ELAUNCH;
            }
            if (zeroed.size() > 64) zeroed.clear();
            zeroed.push_back((void*)ws);
        }
    }
    const BwdPlan plan = plan_at(caller_plan ? const_cast<void*>(caller_plan) : ws + merge_bytes(), B, L);
    uint8_t* ex = ws + merge_bytes() + plan_bytes;
    const size_t cap = (size_t)L * U * UNIT_SAMPLES * 8;
    V* qvals = reinterpret_cast<V*>(ex);
    uint16_t* qkeys = reinterpret_cast<uint16_t*>(ex + up(cap * sizeof(V)));
    const T* g = (const T*)gT;
    T* ge = (T*)gemb;
    if (!caller_plan) bwd_plan<T>(inputs, offsets, B, L, a, plan);
#ifdef LAE_GRID_BWD_PHASE_PROBE
    // probe builds only (tools/fill_beside_mlp_probe.py): LAE_GRID_BWD_PHASE=1 launches the fill pass alone, =2 the accumulate
    // pass alone (on whatever the queue holds); results are meaningless, durations are what is measured
    const char* phase_env = getenv("LAE_GRID_BWD_PHASE");
    const int phase_probe = phase_env ? atoi(phase_env) : 0;
    if (phase_probe != 2)
#endif
    if (a.gridtype == 0 && !a.align && a.interp == 0)
        k_bwd_walk<T, true, true><<<U / SEGS * L, FILL_THREADS, 0, a.stream>>>(g, inputs, offsets, B, L, a.sc, a.gridtype, a.align, a.interp, U, plan, qvals, qkeys, nullptr, nullptr);
    else
        k_bwd_walk<T, true, false><<<U / SEGS * L, FILL_THREADS, 0, a.stream>>>(g, inputs, offsets, B, L, a.sc, a.gridtype, a.align, a.interp, U, plan, qvals, qkeys, nullptr, nullptr);
#ifdef LAE_GRID_BWD_PHASE_PROBE
    if (phase_probe == 1) return lae::check_launch("grid backward (fill-only probe)");
#endif
    // workgroups per level: 32 up to ~400 k samples, 64 beyond (same-box A/B in round 4: lego, 258 k samples, 0.3508 / 0.3547 ms per
    // step at 32 / 64; flower, 658 k, 0.666 / 0.647) -- the smallest levels' few cells take every sample's LDS atomics
    static const uint32_t bk_env = [] { const char* e = getenv("LAE_GRID_BWD_BK_TARGET"); return e ? (uint32_t)std::min(std::max(atoi(e), 1), (int)SUB_RECS) : 0u; }();
    // a split level (P < bk_target) owns one merge record per partition and its arrival tickets are indexed with SUB_RECS: the
    // choice made here may never exceed it.  The records are sized for MAX_LEVELS x SUB_RECS whatever the call
    // (135 MB, zeroed once per buffer: their place must not move with B or L) -- 0.05 % of the card's HBM, a conscious choice.
    static_assert(BK_TARGET <= SUB_RECS && 64u <= SUB_RECS, "bk_target choices must fit the merge records");
    const uint32_t bk_target = std::min(bk_env ? bk_env : (B >= 400000u ? 64u : BK_TARGET), SUB_RECS);
    k_bwd_acc<T><<<(uint32_t)lae::num_cus() * 2, ACC_THREADS, 0, a.stream>>>(offsets, ge, L, a.sc, a.gridtype, a.align, plan, qvals, qkeys, partials, a.nf_flag, bk_target);
    // levels with more partitions than a directory row holds (more than 2^21 entries): generic atomic kernel.  With the
    // caller's host copy of the level sizes the launch is skipped when no level needs it; without one it is always made
    // (its blocks return at once for the levels the binned path has handled).
    bool need_generic = true;
    if (a.offsets_host) {
        need_generic = false;
        for (uint32_t l = 0; l < L; l++)
            if (!level_is_binned((uint32_t)(a.offsets_host[l + 1] - a.offsets_host[l]))) need_generic = true;
    }
    if (need_generic) {
        const uint32_t nbg = lae::cdiv(B, GRID_BLOCK);
        k_grid_bwd<T, 3, 2><<<nbg * L, GRID_BLOCK, 0, a.stream>>>(g, inputs, offsets, ge, B, L, a.sc, a.gridtype, a.align, a.interp, nbg,
                                                                  (L % 8) == 0, 2, (uint64_t)B * 2, true);
    }
    return LAE_OK;
}

// words of the "ever touched" bitmap proper (one bit per 8 entries, + 2 words of slack); MAX_LEVELS words of per-level
// "fully marked" flags follow it (include/laenerf.h)
static inline size_t touched_words(const int32_t* offsets_host, uint32_t L) { return ((size_t)offsets_host[L] >> 3 >> 5) + 2; }

// 0 = binned LDS pipeline where available (default), 1 = always the generic global-atomic kernel
static int g_force_atomic_bwd = 0;

static int grid_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                         void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                         const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                         int dtype, bool blc, void* stream, float in_shift = 0.0f, float in_scale = 1.0f,
                         const int32_t* offsets_host = nullptr, const void* plan = nullptr, int32_t* nf_flag = nullptr,
                         uint32_t* touched = nullptr) {
    (void)embeddings;
    if (B == 0) return LAE_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings) return LAE_ENULL;
    if (plan && !(D == 3 && C == 2 && L <= 32 && B <= BWD_MAX_SAMPLES && !blc && !g_force_atomic_bwd)) return LAE_EINVAL;
    if (gridtype > 1 || interp > 1) return LAE_EINVAL;
    if (touched && dtype != LAE_F16) return LAE_EINVAL;
    if (nf_flag || touched) {
        // only the binned pipeline knows what it stores (the generic kernel's float atomics do not): every level must go
        // through it, which the host copy of the level sizes proves
        if (!(D == 3 && C == 2 && L <= 32 && B <= BWD_MAX_SAMPLES && !g_force_atomic_bwd) || !offsets_host) return LAE_EINVAL;
        for (uint32_t l = 0; l < L; l++)
            if (!level_is_binned((uint32_t)(offsets_host[l + 1] - offsets_host[l]))) return LAE_EINVAL;
    }
    BwdArgs a;
    a.offsets_host = offsets_host;
    a.nf_flag = nf_flag;
    a.touched = touched;
    if (touched) a.touched_full = touched + touched_words(offsets_host, L);
    a.grad = grad; a.inputs = inputs; a.offsets = offsets; a.gemb = grad_embeddings; a.B = B; a.L = L;
    int rc = fill_scales(a.sc, L, S, H);
    if (rc) return rc;
    a.sc.in_shift = in_shift; a.sc.in_scale = in_scale;
    a.gridtype = gridtype; a.align = align_corners != 0; a.interp = interp;
    a.gs_b = blc ? (uint64_t)L * C : C;
    a.gs_l = blc ? C : (uint64_t)B * C;
    a.stream = reinterpret_cast<hipStream_t>(stream);
    if (dtype != LAE_F32 && dtype != LAE_F16) return LAE_EINVAL;
    if (D == 3 && C == 2 && L <= 32 && B <= BWD_MAX_SAMPLES && !g_force_atomic_bwd) {
        // binned path (k_bwd_fill / k_bwd_acc); [B, L*2] grads are transposed into the workspace first
        const size_t esz = dtype == LAE_F16 ? 2 : 4;
        const void* gT = grad;
        if (blc) {
            void* ws = lae::workspace(lae::WS_GRID_GRAD_T, (size_t)B * L * 2 * esz, a.stream);
            if (!ws) return LAE_ELAUNCH;
            if (dtype == LAE_F16) k_grad_transpose<half_t><<<lae::cdiv(B, 256), 256, 0, a.stream>>>((const half_t*)grad, (half_t*)ws, B, L);
            else k_grad_transpose<float><<<lae::cdiv(B, 256), 256, 0, a.stream>>>((const float*)grad, (float*)ws, B, L);
            gT = ws;
        }
        rc = (dtype == LAE_F16) ? launch_bwd_fast<half_t>(gT, inputs, offsets, grad_embeddings, B, L, a, plan)
                                : launch_bwd_fast<float>(gT, inputs, offsets, grad_embeddings, B, L, a, plan);
        if (rc) return rc;
        rc = LAE_OK;
    } else if (dtype == LAE_F32) rc = dispatch_bwd_d<float>(a, D, C);
    else rc = dispatch_bwd_d<half_t>(a, D, C);
    if (rc) return rc;
    if (dy_dx && grad_inputs) {                           // :410 kernel_input_backward
        const uint32_t n = B * D;
        if (dtype == LAE_F32)
            k_grid_input_bwd<float><<<lae::cdiv(n, 256), 256, 0, a.stream>>>((const float*)grad, (const float*)dy_dx,
                                                                            (float*)grad_inputs, B, D, C, L, a.gs_b, a.gs_l);
        else
            k_grid_input_bwd<half_t><<<lae::cdiv(n, 256), 256, 0, a.stream>>>((const half_t*)grad, (const half_t*)dy_dx,
                                                                             (half_t*)grad_inputs, B, D, C, L, a.gs_b, a.gs_l);
    }
    return lae::check_launch("grid_encode_backward");
}

template <int D>
static int tv_c(const float* inputs, const float* emb, float* grad, const int32_t* offsets, float weight, uint32_t B,
                uint32_t C, uint32_t L, const LevelScales& sc, uint32_t gridtype, bool align, hipStream_t s) {
    const uint32_t nb = lae::cdiv(B, GRID_BLOCK);
    switch (C) {
        case 1: k_grad_tv<D, 1><<<nb * L, GRID_BLOCK, 0, s>>>(inputs, emb, grad, offsets, weight, B, L, sc, gridtype, align, nb); break;
        case 2: k_grad_tv<D, 2><<<nb * L, GRID_BLOCK, 0, s>>>(inputs, emb, grad, offsets, weight, B, L, sc, gridtype, align, nb); break;
        case 4: k_grad_tv<D, 4><<<nb * L, GRID_BLOCK, 0, s>>>(inputs, emb, grad, offsets, weight, B, L, sc, gridtype, align, nb); break;
        case 8: k_grad_tv<D, 8><<<nb * L, GRID_BLOCK, 0, s>>>(inputs, emb, grad, offsets, weight, B, L, sc, gridtype, align, nb); break;
        default: return LAE_EINVAL;
    }
    return LAE_OK;
}

}  // namespace

// frame loop (frame.hip lae_render_frame): fp16 table, D=3, C=2, level-major output [L, B_cap, 2]; rows beyond
// *B_dev are not touched.  B_launch = host upper bound of *B_dev (sizes the launch only).
int lae::grid_forward_frame(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B_cap,
                            uint32_t B_launch, const uint32_t* B_dev, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                            int align_corners, uint32_t interp, float in_shift, float in_scale, hipStream_t stream,
                            const int32_t* offsets_host, uint32_t B_likely) {
    if (B_launch == 0) return LAE_OK;
    FwdArgs a;
    a.offsets_host = offsets_host;
    a.inputs = inputs; a.emb = embeddings; a.offsets = offsets; a.out = outputs; a.B = B_cap; a.L = L;
    int rc = fill_scales(a.sc, L, S, H);
    if (rc) return rc;
    a.sc.in_shift = in_shift; a.sc.in_scale = in_scale;
    a.dy_dx = nullptr; a.gridtype = gridtype; a.align = align_corners != 0; a.interp = interp;
    a.stream = stream; a.os_b = 2; a.os_l = (uint64_t)B_cap * 2;
    a.B_dev = B_dev; a.B_launch = std::min(B_launch, B_cap); a.B_likely = std::min(B_likely, a.B_launch);
    launch_fwd<half_t, 3, 2>(a);
    return LAE_OK;
}

extern "C" {

int lae_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                            uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void* stream) {
    return grid_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp,
                        dtype, false, stream);
}
int lae_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                             void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                             uint32_t interp, int dtype, void* stream) {
    return grid_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp, dtype, false, stream);
}

int lae_grid_encode_forward_ex(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                               uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int blc, float in_shift,
                               float in_scale, const int32_t* offsets_host, void* stream) {
    return grid_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp,
                        dtype, blc != 0, stream, in_shift, in_scale, offsets_host);
}
int lae_grid_encode_backward_ex(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                                int align_corners, uint32_t interp, int dtype, int blc, float in_shift, float in_scale,
                                const int32_t* offsets_host, int32_t* nonfinite_flag, uint32_t* touched_lines, void* stream) {
    return grid_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp, dtype, blc != 0, stream, in_shift, in_scale, offsets_host, nullptr,
                         nonfinite_flag, touched_lines);
}

uint64_t lae_grid_backward_workspace_bytes(uint32_t B, uint32_t L, int dtype) {
    return plan_layout(B, L).bytes + (dtype == LAE_F16 ? bwd_exec_ws_bytes<half_t>(B, L) : bwd_exec_ws_bytes<float>(B, L));
}

uint64_t lae_grid_backward_plan_bytes(uint32_t B, uint32_t L) { return plan_layout(B, L).bytes; }

uint64_t lae_grid_touched_lines_words(uint64_t n_entries) { return (n_entries >> 3 >> 5) + 2 + MAX_LEVELS; }

int lae_grid_encode_backward_plan(const float* inputs, const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                  uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float in_shift,
                                  float in_scale, const int32_t* offsets_host, void* plan, uint32_t* touched_lines, void* stream) {
    if (B == 0) return LAE_OK;
    if (!inputs || !offsets || !plan) return LAE_ENULL;
    if (D != 3 || C != 2 || L > 32 || B > BWD_MAX_SAMPLES || gridtype > 1 || interp > 1 || (dtype != LAE_F32 && dtype != LAE_F16)) return LAE_EINVAL;
    if (touched_lines) {                                   // every level must go through this pass (see lae_grid_encode_backward_ex)
        if (!offsets_host || dtype != LAE_F16 || g_force_atomic_bwd) return LAE_EINVAL;
        for (uint32_t l = 0; l < L; l++)
            if (!level_is_binned((uint32_t)(offsets_host[l + 1] - offsets_host[l]))) return LAE_EINVAL;
    }
    BwdArgs a;
    a.offsets_host = offsets_host;
    a.touched = touched_lines;
    if (touched_lines) a.touched_full = touched_lines + touched_words(offsets_host, L);
    a.grad = nullptr; a.inputs = inputs; a.offsets = offsets; a.gemb = nullptr; a.B = B; a.L = L;
    int rc = fill_scales(a.sc, L, S, H);
    if (rc) return rc;
    a.sc.in_shift = in_shift; a.sc.in_scale = in_scale;
    a.gridtype = gridtype; a.align = align_corners != 0; a.interp = interp;
    a.gs_b = C; a.gs_l = (uint64_t)B * C;
    a.stream = reinterpret_cast<hipStream_t>(stream);
    if (dtype == LAE_F16) bwd_plan<half_t>(inputs, offsets, B, L, a, plan_at(plan, B, L));
    else bwd_plan<float>(inputs, offsets, B, L, a, plan_at(plan, B, L));
    return lae::check_launch("grid_encode_backward_plan");
}

int lae_grid_encode_backward_planned(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                     uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                     uint32_t interp, int dtype, float in_shift, float in_scale, const int32_t* offsets_host,
                                     const void* plan, int32_t* nonfinite_flag, void* stream) {
    if (!plan) return LAE_ENULL;
    return grid_backward(grad, inputs, nullptr, offsets, grad_embeddings, B, D, C, L, S, H, nullptr, nullptr, gridtype, align_corners,
                         interp, dtype, false, stream, in_shift, in_scale, offsets_host, plan, nonfinite_flag, nullptr);
}

int lae_grid_forward_schedule(const int32_t* offsets_host, uint32_t L, float S, uint32_t H, uint32_t n_chunks, uint32_t* nseg_out,
                              uint32_t* segs_out) {
    if (!offsets_host || !nseg_out || !segs_out) return LAE_ENULL;
    LevelScales sc;
    if (fill_scales(sc, L, S, H) != LAE_OK || L > 8 * FWD_MAX_SEG) return LAE_EINVAL;
    FwdSched fs;
    const uint32_t per_xcd = fwd_sched_build(fs, L, n_chunks, sc, offsets_host);
    for (int x = 0; x < 8; x++) {
        nseg_out[x] = fs.nseg[x];
        for (uint32_t q = 0; q < fs.nseg[x]; q++) {
            uint32_t* o = segs_out + ((size_t)x * FWD_MAX_SEG + q) * 3;
            o[0] = fs.seg[x][q].level; o[1] = fs.seg[x][q].c0; o[2] = fs.seg[x][q].n;
        }
    }
    return (int)per_xcd;
}

int lae_grid_set_forward_mode(int mode) {
    if (mode < 0 || mode > 2) return LAE_EINVAL;
    g_fwd_mode = mode;
    return LAE_OK;
}

int lae_grid_set_backward_mode(int mode) {
    if (mode != 0 && mode != 1) return LAE_EINVAL;
    g_force_atomic_bwd = mode;
    return LAE_OK;
}

int lae_grad_total_variation(const void* inputs, const void* embeddings, void* grad, const int32_t* offsets,
                             float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             uint32_t gridtype, int align_corners, int dtype, void* stream) {
    if (B == 0) return LAE_OK;
    if (!inputs || !embeddings || !grad || !offsets) return LAE_ENULL;
    if (dtype != LAE_F32 || gridtype > 1) return LAE_EINVAL;
    LevelScales sc;
    int rc = fill_scales(sc, L, S, H);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool al = align_corners != 0;
    const float* in = (const float*)inputs; const float* e = (const float*)embeddings; float* g = (float*)grad;
    switch (D) {
        case 2: rc = tv_c<2>(in, e, g, offsets, weight, B, C, L, sc, gridtype, al, s); break;
        case 3: rc = tv_c<3>(in, e, g, offsets, weight, B, C, L, sc, gridtype, al, s); break;
        case 4: rc = tv_c<4>(in, e, g, offsets, weight, B, C, L, sc, gridtype, al, s); break;
        case 5: rc = tv_c<5>(in, e, g, offsets, weight, B, C, L, sc, gridtype, al, s); break;
        default: rc = LAE_EINVAL;
    }
    if (rc) return rc;
    return lae::check_launch("grad_total_variation");
}

}  // extern "C"

#ifdef LAE_GRID_STAMPS
extern "C" __attribute__((visibility("default"))) int lae_debug_grid_stamps(void* out, size_t bytes) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grid_stamps), bytes) == hipSuccess ? 0 : 1;
}
#endif
