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

namespace {

constexpr int MAX_LEVELS = 32;
// scale[l] = exp2(l*S)*H - 1 (host, :138).  in_shift / in_scale: optional affine map applied to every coordinate as it
// is read, x01 = (x + in_shift) * in_scale -- GridEncoder.forward's `(inputs + bound) / (2 * bound)` (grid.py:149; torch
// evaluates the division by a Python scalar as a multiplication by its fp32 reciprocal) without two extra kernels.
struct LevelScales { float scale[MAX_LEVELS]; float in_shift, in_scale; };

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__constant__ uint32_t k_primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};

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
struct FwdSeg { uint32_t level, c0, n; };                  // chunks [c0, c0 + n) of `level`
constexpr int FWD_MAX_SEG = 12;
struct FwdSched { uint32_t nseg[8]; FwdSeg seg[8][FWD_MAX_SEG]; };

__global__ __launch_bounds__(GRID_BLOCK) void k_grid_fwd_lean(
    const float* __restrict__ inputs, const half_t* __restrict__ grid, const int32_t* __restrict__ offsets,
    half_t* __restrict__ outputs, uint32_t B, LevelScales sc, FwdSched sched, uint64_t os_b, uint64_t os_l,
    const uint32_t* __restrict__ B_dev) {
    const uint32_t xcd = blockIdx.x & 7u;
    uint32_t j = blockIdx.x >> 3, level = 0xffffffffu, chunk = 0;
    const uint32_t ns = sched.nseg[xcd];
    for (uint32_t q = 0; q < ns; q++) {
        const uint32_t n = sched.seg[xcd][q].n;
        if (j < n) { level = sched.seg[xcd][q].level; chunk = sched.seg[xcd][q].c0 + j; break; }
        j -= n;
    }
    if (level == 0xffffffffu) return;
    const uint32_t b = chunk * GRID_BLOCK + threadIdx.x;
    if (B_dev) B = min(B, *B_dev);
    if (b >= B) return;
    const LevelInfo<3> li = level_info<3>(sc, offsets, level, 0u, false);
    const char* __restrict__ tabb = reinterpret_cast<const char*>(grid + (size_t)li.table_off * 2);

    struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
    const F3 in = *reinterpret_cast<const F3*>(inputs + (size_t)b * 3);
    const float xin[3] = {in.x, in.y, in.z};
    float fr[3];
    uint32_t pg[3];
    bool oob = false;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float x01 = (xin[d] + sc.in_shift) * sc.in_scale;
        oob |= (x01 < 0.0f) | (x01 > 1.0f);
        const float p = fmaf(x01, li.scale, 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)fl;
        fr[d] = p - (float)pg[d];
    }
    half2_t* out = reinterpret_cast<half2_t*>(outputs + (size_t)b * os_b + (size_t)level * os_l);
    if (oob) { const half2_t z = {(half_t)0.0f, (half_t)0.0f}; *out = z; return; }   // gridencoder.cu:118-135

    uint32_t cw[8];                                        // corner idx = x + 2y + 4z, two halves each
    if (li.use_hash && li.pow2) {
        const uint32_t m4 = (li.hashmap_size - 1u) << 2;
        const uint32_t hy0 = (pg[1] * 2654435761u) << 2, hy1 = hy0 + (2654435761u << 2);
        const uint32_t hz0 = (pg[2] * 805459861u) << 2, hz1 = hz0 + (805459861u << 2);
        const uint32_t x0 = pg[0] << 2, x1 = x0 + 4u;
        const uint32_t h[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        if ((pg[0] & 1u) == 0 && li.hashmap_size >= 2) {   // idx(x + 1) == idx(x) ^ 1: one aligned 8-byte load per row
#pragma unroll
            for (int yz = 0; yz < 4; yz++) {
                const uint32_t o0 = (x0 ^ h[yz]) & m4;
                const uint2 w = *reinterpret_cast<const uint2*>(tabb + (o0 & ~4u));
                cw[2 * yz] = (o0 & 4u) ? w.y : w.x;
                cw[2 * yz + 1] = (o0 & 4u) ? w.x : w.y;
            }
        } else {
#pragma unroll
            for (int yz = 0; yz < 4; yz++) {
                cw[2 * yz] = *reinterpret_cast<const uint32_t*>(tabb + ((x0 ^ h[yz]) & m4));
                cw[2 * yz + 1] = *reinterpret_cast<const uint32_t*>(tabb + ((x1 ^ h[yz]) & m4));
            }
        }
    } else if (!li.use_hash && li.nowrap) {
        const uint32_t base = (pg[0] + pg[1] * li.stride[1] + pg[2] * li.stride[2]) << 2;     // stride[0] == 1
        const uint32_t dy = li.stride[1] << 2, dz = li.stride[2] << 2;
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            const uint32_t o0 = base + ((yz & 1) ? dy : 0u) + ((yz >> 1) ? dz : 0u);
            cw[2 * yz] = *reinterpret_cast<const uint32_t*>(tabb + o0);
            cw[2 * yz + 1] = *reinterpret_cast<const uint32_t*>(tabb + o0 + 4u);
        }
    } else {
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const uint32_t pl[3] = {pg[0] + (idx & 1), pg[1] + ((idx >> 1) & 1), pg[2] + (idx >> 2)};
            cw[idx] = *reinterpret_cast<const uint32_t*>(tabb + ((size_t)cell_index<3>(li, pl) << 2));
        }
    }
    half_t r0 = (half_t)0.0f, r1 = (half_t)0.0f;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        const float w = (((idx & 1) ? fr[0] : 1 - fr[0]) * ((idx & 2) ? fr[1] : 1 - fr[1])) * ((idx & 4) ? fr[2] : 1 - fr[2]);
        const half2_t v = __builtin_bit_cast(half2_t, cw[idx]);
        accum(r0, w, v[0]);
        accum(r1, w, v[1]);
    }
    const half2_t h2 = {r0, r1};
    *out = h2;
}

// ---------------------------------------------------------------- K14
// gridencoder.cu:248-340: scatter w*grad into grad_grid.  v1: one no-return atomic per
// (corner, channel pair): global_atomic_add_f32 / global_atomic_pk_add_f16.
// ---- constants and predicates of the binned backward (defined here because the generic kernel below is its companion)
constexpr uint32_t BK_MAX = 256;              // buckets per level in the tables
constexpr uint32_t BK_TARGET = 16;            // target workgroups per level
constexpr int BIN_THREADS = 256;
constexpr int BIN_SPT = 16;                   // consecutive samples per lane
constexpr uint32_t E_NONE = 0xffffu;

template <typename T> struct HShift { static constexpr uint32_t value = sizeof(T) == 2 ? 13 : 14; };
template <typename T> struct HItem;
template <> struct HItem<half_t> { uint32_t e; half2_t v0, v1; uint32_t pad; };            // 16 B
template <> struct HItem<float> { uint32_t e; float v0x, v0y, v1x, v1y; uint32_t pad; };    // 24 B

struct LevelBins { uint32_t P, SUB; };
template <typename T>
__host__ __device__ __forceinline__ LevelBins level_bins_of(uint32_t hashmap_size) {
    constexpr uint32_t SHIFT = HShift<T>::value;
    LevelBins lb;
    lb.P = (hashmap_size + (1u << SHIFT) - 1) >> SHIFT;
    lb.SUB = lb.P >= BK_TARGET ? 1u : (BK_TARGET + lb.P - 1) / lb.P;
    return lb;
}
template <typename T>
__device__ __forceinline__ LevelBins level_bins(const LevelInfo<3>& li) { return level_bins_of<T>(li.hashmap_size); }

// A level goes through the binned pipeline when its buckets fit the tables and, for hashed levels, both x corners of a
// (y', z') pair provably share a partition (x' + 1 < 2^SHIFT).  Everything else is left to the generic atomic kernel.
template <typename T>
__host__ __device__ __forceinline__ bool level_is_binned(uint32_t hashmap_size, uint32_t resolution, bool use_hash) {
    const LevelBins lb = level_bins_of<T>(hashmap_size);
    return lb.P * lb.SUB <= BK_MAX && !(use_hash && resolution + 1 > (1u << HShift<T>::value));
}


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
        if constexpr (D == 3) { if (level_is_binned<T>(li.hashmap_size, li.resolution, li.use_hash)) return; }
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

// ---------------------------------------------------------------- K14, MI355X form (D = 3, C = 2)
// Scattered global float atomics execute at the memory side on gfx950 (~20 G requests/s chip-wide,
// MI355X_MICROARCH "Global float atomics"): 128 of them per sample made k_grid_bwd 54% of the train step.
// Here the gradient table is cut into 128 KiB partitions (32768 half2 / 16384 float2 entries) and ONE
// persistent workgroup owns a partition in LDS: it streams the samples, keeps the contributions that fall
// into its partition with LDS atomics (ds_pk_add_f16 / ds_add_f32) and finally adds the partition to
// grad_grid with plain coalesced read-modify-writes.  No global atomic on the hashed levels.
//   hashed level : partition = index >> SHIFT depends only on the (y', z') corner pair because x' < 2^SHIFT, so
//                  one test covers the two x corners; every partition workgroup scans all samples.
//   dense level  : few partitions; the samples are additionally cut into slices and each LANE walks a run of
//                  consecutive samples, summing in registers while the cell stays the same (samples of one ray
//                  share coarse cells) -- this removes the same-address LDS conflicts.  Slices of one partition
//                  are combined with contiguous (full-rate) global atomics.
// grads come in [L][B][2] (the reference's layout); the [B, L*2] variant is transposed into a workspace first.
constexpr int LB_THREADS = 1024;

template <typename T>
__device__ __forceinline__ void lds_acc_add(uint32_t* acc, uint32_t e, float v0, float v1) {
    if constexpr (sizeof(T) == 2) {
        half2_t v = {(half_t)v0, (half_t)v1};
        __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(acc) + e, v);
    } else {
        float* a = reinterpret_cast<float*>(acc) + 2 * e;
        atomicAdd(a, v0);
        atomicAdd(a + 1, v1);
    }
}

// MI355X grid backward, work-efficient form (D = 3, C = 2).
//   k_bin<COUNT> : every lane walks BIN_SPT CONSECUTIVE samples of one level (samples of a ray stay in a cell for
//                  many steps at coarse/mid levels) and sums the 8 corner contributions in registers while the cell is
//                  unchanged; each finished cell emits ITEMS = {entry offsets e0|e1<<16, value0, value1}:
//                    hashed level : 4 items, one per (y',z') corner pair (both x corners share a partition because
//                                   x' < 2^SHIFT, and bucket = hash >> SHIFT depends on (y',z') only)
//                    dense level  : 8 single-corner items (e1 = NONE)
//                  COUNT pass: per-bucket item counts (LDS histogram -> one global add per bucket per block).
//   k_bin_scan   : exclusive scan of the bucket counts -> queue offsets / cursors.
//   k_bin<FILL>  : same walk, items appended to their bucket's queue (block-local rank from the LDS histogram,
//                  one global cursor add per bucket per block -> contiguous runs).
//   k_bin_acc    : one workgroup per bucket: queue -> LDS accumulators -> grad_grid.
// bucket = (level, partition p of 2^SHIFT entries, sub-bucket): levels with few partitions are split into SUB
// sub-buckets by block id so that ~16 workgroups share every level.
// fp16 grads accumulate EXACTLY as 2^-24 fixed point in int64 LDS words (ds_add_u64 ~0.8 cycles / lane-op vs ~3.2 for
// ds_add_f32 / ds_pk_add_f16 on gfx950, tools/ubench/lds_atomic.hip; every fp16 value is a multiple of 2^-24), so the
// result is the correctly rounded sum of the fp16 contributions, independent of order.  fp32 grads use ds_add_f32.
template <typename T, bool FILL>
__global__ __launch_bounds__(BIN_THREADS) void k_bin(
    const T* __restrict__ gradT, const float* __restrict__ inputs, const int32_t* __restrict__ offsets, uint32_t B,
    uint32_t L, LevelScales sc, uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t nb,
    uint32_t* __restrict__ block_counts, const uint32_t* __restrict__ offs, HItem<T>* __restrict__ queue) {
    constexpr uint32_t SHIFT = HShift<T>::value, PART = 1u << SHIFT;
    const uint32_t level = blockIdx.x / nb, chunk = blockIdx.x % nb;
    const LevelInfo<3> li = level_info<3>(sc, offsets, level, gridtype, align_corners);
    const LevelBins lb = level_bins<T>(li);
    const uint32_t nbk = lb.P * lb.SUB;
    if (!level_is_binned<T>(li.hashmap_size, li.resolution, li.use_hash)) return;   // handled by the generic atomic kernel
    __shared__ uint32_t hist[BK_MAX];
    __shared__ uint32_t base[BK_MAX];
    const uint32_t tid = threadIdx.x;
    uint32_t* __restrict__ my_counts = block_counts + ((size_t)level * nb + chunk) * BK_MAX;
    hist[tid] = 0;                                      // BIN_THREADS == BK_MAX
    // FILL: queue position of this block's first item per bucket = bucket offset + exclusive count of earlier blocks
    if (FILL) base[tid] = tid < nbk ? offs[level * BK_MAX + tid] + my_counts[tid] : 0u;
    __syncthreads();
    const uint32_t sub = chunk % lb.SUB;
    const uint32_t pmask = lb.P - 1, emask = min(li.hashmap_size, PART) - 1;
    const bool nowrap = li.nowrap;
    const uint32_t b0 = (chunk * BIN_THREADS + tid) * BIN_SPT;

    // ---- pass A: walk the samples, build the merged cells in registers (at most BIN_SPT cells)
    // per cell: key coords + 16 accumulated values.  Processed twice (rank, then emit) to keep registers small.
    float xs[BIN_SPT][3]; float g0[BIN_SPT], g1[BIN_SPT];
    if (b0 + BIN_SPT <= B) {
        const float4* src = reinterpret_cast<const float4*>(inputs + (size_t)b0 * 3);      // 96 B per lane, 16 B aligned
        float4 v[BIN_SPT * 3 / 4];
#pragma unroll
        for (int i = 0; i < BIN_SPT * 3 / 4; i++) v[i] = src[i];
        const float* vf = reinterpret_cast<const float*>(v);
#pragma unroll
        for (int s_ = 0; s_ < BIN_SPT; s_++) { xs[s_][0] = vf[3 * s_]; xs[s_][1] = vf[3 * s_ + 1]; xs[s_][2] = vf[3 * s_ + 2]; }
    } else {
#pragma unroll
        for (int s_ = 0; s_ < BIN_SPT; s_++) {
            const uint32_t b = min(b0 + s_, B - 1);
            xs[s_][0] = inputs[(size_t)b * 3]; xs[s_][1] = inputs[(size_t)b * 3 + 1]; xs[s_][2] = inputs[(size_t)b * 3 + 2];
        }
    }
#pragma unroll
    for (int s_ = 0; s_ < BIN_SPT; s_++)
#pragma unroll
        for (int d = 0; d < 3; d++) xs[s_][d] = (xs[s_][d] + sc.in_shift) * sc.in_scale;
    if constexpr (FILL) {                               // the COUNT pass depends on the positions only (it can run before
        const T* __restrict__ g_lvl = gradT + (size_t)level * B * 2;   // the gradients exist: lae_grid_encode_backward_plan)
        // a lane's BIN_SPT gradients are contiguous (64 B in fp16): wide loads when the whole window is in range and the
        // level's slice keeps 16-byte alignment (one vector-memory instruction per 4 / 2 samples instead of one each)
        if (b0 + BIN_SPT <= B && ((size_t)level * B * 2 * sizeof(T)) % 16 == 0 && (reinterpret_cast<uintptr_t>(gradT) & 15) == 0) {
            if constexpr (sizeof(T) == 2) {
                const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const half2_t*>(g_lvl) + b0);
#pragma unroll
                for (int q = 0; q < BIN_SPT / 4; q++) {
                    const uint4 v = src[q];
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const half2_t gv = __builtin_bit_cast(half2_t, w[k]);
                        g0[4 * q + k] = (float)gv[0]; g1[4 * q + k] = (float)gv[1];
                    }
                }
            } else {
                const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(g_lvl) + b0);
#pragma unroll
                for (int q = 0; q < BIN_SPT / 2; q++) {
                    const float4 v = src[q];
                    g0[2 * q] = v.x; g1[2 * q] = v.y; g0[2 * q + 1] = v.z; g1[2 * q + 1] = v.w;
                }
            }
        } else {
#pragma unroll
        for (int s_ = 0; s_ < BIN_SPT; s_++) {
            const uint32_t b = min(b0 + s_, B - 1);
            if constexpr (sizeof(T) == 2) { const half2_t gv = reinterpret_cast<const half2_t*>(g_lvl)[b]; g0[s_] = (float)gv[0]; g1[s_] = (float)gv[1]; }
            else { const float2 gv = reinterpret_cast<const float2*>(g_lvl)[b]; g0[s_] = gv.x; g1[s_] = gv.y; }
        }
        }
    } else {
#pragma unroll
        for (int s_ = 0; s_ < BIN_SPT; s_++) { g0[s_] = 0.f; g1[s_] = 0.f; }
    }

    // emit one finished cell: `mode` 0 = count/rank (returns ranks through rk[]), 1 = write items
    uint32_t cpg[3] = {0, 0, 0};
    float a0[8], a1[8];
    auto cell_items = [&](auto&& visit) {
        // visit(bucket, e, v0a, v0b, v1a, v1b) for every item of the current cell
        if (li.use_hash) {
            const uint32_t hy0 = cpg[1] * 2654435761u, hy1 = hy0 + 2654435761u;
            const uint32_t hz0 = cpg[2] * 805459861u, hz1 = hz0 + 805459861u;
#pragma unroll
            for (int yz = 0; yz < 4; yz++) {
                const uint32_t h = ((yz & 1) ? hy1 : hy0) ^ ((yz & 2) ? hz1 : hz0);
                const uint32_t bk = ((h >> SHIFT) & pmask) * lb.SUB + sub;
                const uint32_t e = ((cpg[0] ^ h) & emask) | ((((cpg[0] + 1) ^ h) & emask) << 16);
                visit(yz, bk, e, a0[2 * yz], a1[2 * yz], a0[2 * yz + 1], a1[2 * yz + 1]);
            }
        } else {
            const uint32_t key = cpg[0] * li.stride[0] + cpg[1] * li.stride[1] + cpg[2] * li.stride[2];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                uint32_t idx = key + ((c & 1) ? li.stride[0] : 0u) + ((c & 2) ? li.stride[1] : 0u) + ((c & 4) ? li.stride[2] : 0u);
                if (!nowrap) idx = li.pow2 ? (idx & (li.hashmap_size - 1)) : (idx % li.hashmap_size);
                const uint32_t bk = (idx >> SHIFT) * lb.SUB + sub;
                // corner order c = x + 2y + 4z; accumulators are stored pair-major: slot = 2*(c>>1) + (c&1)
                visit(c, bk, (idx & (PART - 1)) | (E_NONE << 16), a0[c], a1[c], 0.0f, 0.0f);
            }
        }
    };

    // one sweep over the lane's samples.  COUNT: histogram only.  FILL: slot = LDS fill counter (any unique slot in
    // the block's reserved range is fine; the accumulation is order independent for fp16, see k_bin_acc).
    {
        bool have = false;
#pragma unroll
        for (int c = 0; c < 8; c++) { a0[c] = 0; a1[c] = 0; }
        auto finish = [&]() {
            if (!FILL) {
                cell_items([&](int, uint32_t bk, uint32_t, float, float, float, float) { atomicAdd(&hist[bk], 1u); });
            } else {
                cell_items([&](int, uint32_t bk, uint32_t e, float va, float vb, float vc, float vd) {
                    HItem<T> it;
                    it.e = e; it.pad = 0;
                    if constexpr (sizeof(T) == 2) { it.v0 = half2_t{(half_t)va, (half_t)vb}; it.v1 = half2_t{(half_t)vc, (half_t)vd}; }
                    else { it.v0x = va; it.v0y = vb; it.v1x = vc; it.v1y = vd; }
                    queue[(size_t)base[bk] + atomicAdd(&hist[bk], 1u)] = it;
                });
            }
#pragma unroll
            for (int c = 0; c < 8; c++) { a0[c] = 0; a1[c] = 0; }
        };
#pragma unroll
        for (int s_ = 0; s_ < BIN_SPT; s_++) {
            if (b0 + s_ >= B) break;
            float frac[3]; uint32_t pg[3]; bool ok = true;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float xv = xs[s_][d];
                ok = ok && !(xv < 0.0f) && !(xv > 1.0f);
                float pp = fmaf(xv, li.scale, align_corners ? 0.0f : 0.5f);
                const float fl = floorf(pp);
                pg[d] = (uint32_t)fl;
                pp -= (float)pg[d];
                if (interp == 1) pp = pp * pp * (3.0f - 2.0f * pp);
                frac[d] = pp;
            }
            if (!ok) continue;
            if (!have || pg[0] != cpg[0] || pg[1] != cpg[1] || pg[2] != cpg[2]) {
                if (have) finish();
                cpg[0] = pg[0]; cpg[1] = pg[1]; cpg[2] = pg[2]; have = true;
            }
            // slot c = x + 2y + 4z = x + 2*yz: the same numbering serves the pair items (hashed) and corner items (dense)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float w = (((c & 1) ? frac[0] : 1 - frac[0]) * ((c & 2) ? frac[1] : 1 - frac[1])) * ((c & 4) ? frac[2] : 1 - frac[2]);
                a0[c] = fmaf(w, g0[s_], a0[c]); a1[c] = fmaf(w, g1[s_], a1[c]);
            }
        }
        if (have) finish();
    }
    if (!FILL) {
        __syncthreads();
        my_counts[tid] = hist[tid];                     // coalesced; zeros included (the scan reads every slot)
    }
}

// per bucket: exclusive scan over the blocks' counts (in place) and bucket total.  One wavefront per (level, bucket).
__global__ __launch_bounds__(256) void k_bin_scan_blocks(uint32_t* __restrict__ block_counts, uint32_t* __restrict__ counts,
                                                          uint32_t L, uint32_t nb) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= L * BK_MAX) return;
    const int lane = threadIdx.x & 63;
    const uint32_t level = t / BK_MAX, bk = t % BK_MAX;
    uint32_t* p = block_counts + (size_t)level * nb * BK_MAX + bk;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t v = c < nb ? p[(size_t)c * BK_MAX] : 0u;
        const uint32_t inc = lae::wave_incl_scan(v);
        if (c < nb) p[(size_t)c * BK_MAX] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) counts[t] = carry;
}

// bucket totals -> exclusive offsets (global item index).  One small block.
__global__ __launch_bounds__(1024) void k_bin_scan(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offs, uint32_t n,
                                                   uint32_t* __restrict__ tickets) {
    __shared__ uint32_t lds[17];
    if (threadIdx.x < 2) tickets[threadIdx.x] = 0;          // work queue of the accumulate pass starts empty-handed
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? counts[i] : 0;
        uint32_t total;
        const uint32_t ex = lae::block_excl_scan<16>(v, &total, lds);
        if (i < n) offs[i] = carry + ex;
        carry += total;
    }
}

// fp16 bits -> value * 2^24 as a signed integer (exact: fp16 values are multiples of 2^-24, |v| * 2^24 < 2^40)
__device__ __forceinline__ long long half_to_fix24(half_t v) {
    const uint32_t u = (uint32_t)__builtin_bit_cast(uint16_t, v);
    const uint32_t e = (u >> 10) & 31u, m = u & 1023u;
    const unsigned long long mag = e ? ((unsigned long long)(m | 1024u) << (e - 1)) : (unsigned long long)m;
    return (u & 0x8000u) ? -(long long)mag : (long long)mag;
}

template <typename T>
__global__ __launch_bounds__(LB_THREADS) void k_bin_acc(
    const int32_t* __restrict__ offsets, T* __restrict__ grad_grid, uint32_t L, LevelScales sc, uint32_t gridtype,
    bool align_corners, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offs,
    const HItem<T>* __restrict__ queue, uint32_t* __restrict__ tickets) {
    constexpr bool HALF = sizeof(T) == 2;
    constexpr uint32_t SHIFT = HShift<T>::value, PART = 1u << SHIFT;
    __shared__ unsigned long long acc64[16384];         // 128 KiB: int64[8192][2] (fp16 grads) or float2[16384]
    const uint32_t tid = threadIdx.x;
    // bucket list, level-major; per-level bucket counts computed ONCE into LDS (decoding must not walk offsets[]
    // with dependent global loads: that serial walk cost ~300 us in an earlier version)
    __shared__ uint32_t s_cnt[MAX_LEVELS], s_sub[MAX_LEVELS];
    if (tid < L) {
        const LevelInfo<3> li_ = level_info<3>(sc, offsets, tid, gridtype, align_corners);
        const LevelBins lb = level_bins<T>(li_);
        s_cnt[tid] = level_is_binned<T>(li_.hashmap_size, li_.resolution, li_.use_hash) ? lb.P * lb.SUB : 0u;
        s_sub[tid] = lb.SUB;
    }
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t l = 0; l < L; l++) total += s_cnt[l];
    // Buckets are handed out through one device-side counter, finest levels (the full queues) first: with the static
    // round-robin deal 788 buckets over 256 workgroups meant 3 buckets for most and 4 for some, of very different sizes.
    __shared__ uint32_t s_ticket;
    for (;;) {
        if (tid == 0) s_ticket = atomicAdd(&tickets[0], 1u);
        __syncthreads();
        const uint32_t t = s_ticket;
        __syncthreads();
        if (t >= total) break;
        const uint32_t item = total - 1u - t;
        uint32_t level = 0, bk = item;
        while (bk >= s_cnt[level]) { bk -= s_cnt[level]; level++; }
        const uint32_t SUB = s_sub[level], p = bk / SUB;
        const LevelInfo<3> li = level_info<3>(sc, offsets, level, gridtype, align_corners);
        const uint32_t n = counts[level * BK_MAX + bk];
        if (n == 0) continue;                               // uniform per block
        for (uint32_t i = tid; i < 16384; i += LB_THREADS) acc64[i] = 0ull;
        __syncthreads();
        const HItem<T>* __restrict__ q = queue + offs[level * BK_MAX + bk];
        auto apply = [&](const HItem<T>& it) {
            const uint32_t e0 = it.e & 0xffffu, e1 = it.e >> 16;
            if constexpr (HALF) {
                atomicAdd(&acc64[2 * e0], (unsigned long long)half_to_fix24(it.v0[0]));
                atomicAdd(&acc64[2 * e0 + 1], (unsigned long long)half_to_fix24(it.v0[1]));
                if (e1 != E_NONE) {
                    atomicAdd(&acc64[2 * e1], (unsigned long long)half_to_fix24(it.v1[0]));
                    atomicAdd(&acc64[2 * e1 + 1], (unsigned long long)half_to_fix24(it.v1[1]));
                }
            } else {
                float* af = reinterpret_cast<float*>(acc64);
                atomicAdd(af + 2 * e0, it.v0x); atomicAdd(af + 2 * e0 + 1, it.v0y);
                if (e1 != E_NONE) { atomicAdd(af + 2 * e1, it.v1x); atomicAdd(af + 2 * e1 + 1, it.v1y); }
            }
        };
        // four queue loads in flight per lane before the first LDS atomic (the loop is latency bound otherwise)
        uint32_t i = tid;
        for (; i + 3 * LB_THREADS < n; i += 4 * LB_THREADS) {
            const HItem<T> i0 = q[i], i1 = q[i + LB_THREADS], i2 = q[i + 2 * LB_THREADS], i3 = q[i + 3 * LB_THREADS];
            apply(i0); apply(i1); apply(i2); apply(i3);
        }
        for (; i < n; i += LB_THREADS) apply(q[i]);
        __syncthreads();
        const uint32_t part_lo = p << SHIFT;
        const uint32_t n_ent = min(PART, li.hashmap_size - part_lo);
        T* __restrict__ dst = grad_grid + ((size_t)li.table_off + part_lo) * 2;
        if constexpr (HALF) {
            half2_t* d2 = reinterpret_cast<half2_t*>(dst);
            for (uint32_t e = tid; e < n_ent; e += LB_THREADS) {
                const long long i0 = (long long)acc64[2 * e], i1 = (long long)acc64[2 * e + 1];
                const float s0 = (float)i0 * 5.9604644775390625e-08f, s1 = (float)i1 * 5.9604644775390625e-08f;   // * 2^-24
                if (SUB == 1) {      // only writer of this table slice: plain coalesced read-modify-write
                    const half2_t o = d2[e];
                    d2[e] = half2_t{(half_t)((float)o[0] + s0), (half_t)((float)o[1] + s1)};
                } else if (i0 != 0 || i1 != 0) {
                    __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t*)(d2 + e), half2_t{(half_t)s0, (half_t)s1});
                }
            }
        } else {
            const float* af = reinterpret_cast<const float*>(acc64);
            for (uint32_t e = tid; e < 2 * n_ent; e += LB_THREADS) {
                const float v = af[e];
                if (SUB == 1) dst[e] += v; else if (v != 0.0f) atomicAdd(dst + e, v);
            }
        }
        __syncthreads();
    }
    if (tid == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1) { tickets[0] = 0; tickets[1] = 0; }   // last one out resets
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
};

// ---- schedule of k_grid_fwd_lean (see the kernel's header).  Relative cost of one chunk of a level, calibrated on the
// bench batches (tools/ubench/grid_fwd_variants.hip, then a sweep with tools/grid_fwd_bench.py on one-view and 16-view
// batches): dense 1; hashed 1.25 up to resolution ~80, rising with log2(resolution) to 2.3 at ~550 (consecutive samples of a
// ray stop sharing cache lines) and 4.5 from ~1000 on (every corner row is its own L2 request).  Only the balance depends on it, never a result.
static const std::vector<int32_t>* host_offsets(const int32_t* offsets, uint32_t L, hipStream_t stream);
static float fwd_level_cost(bool hashed, uint32_t resolution) {
    if (!hashed) return 1.0f;
    const float lr = log2f((float)resolution);
    if (lr <= 6.3f) return 1.25f;
    if (lr <= 9.1f) return 1.25f + (lr - 6.3f) * (1.05f / 2.8f);
    if (lr <= 10.0f) return 2.3f + (lr - 9.1f) * (2.2f / 0.9f);
    return 4.5f;
}
static void fwd_sched_default(FwdSched& fs, uint32_t L, uint32_t nb) {     // level l on XCD l mod 8, one level at a time
    for (int x = 0; x < 8; x++) fs.nseg[x] = 0;
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t x = l & 7u;
        fs.seg[x][fs.nseg[x]++] = FwdSeg{l, 0u, nb};
    }
}
// returns the largest number of blocks any XCD owns
static uint32_t fwd_sched_build(FwdSched& fs, uint32_t L, uint32_t nb, const LevelScales& sc, const std::vector<int32_t>* offs) {
    bool ok = offs != nullptr && L <= 32 && nb >= 64;
    float cost[MAX_LEVELS]; bool dense[MAX_LEVELS];
    if (ok) {
        for (uint32_t l = 0; l < L; l++) {
            const uint32_t res = (uint32_t)ceilf(sc.scale[l]) + 1;
            const uint64_t size = (uint64_t)((*offs)[l + 1] - (*offs)[l]);
            const uint64_t full = (uint64_t)(res + 1) * (res + 1) * (res + 1);
            dense[l] = full <= size;
            cost[l] = fwd_level_cost(!dense[l], res);
        }
        struct Item { float cost; uint32_t level; bool piece; };
        std::vector<Item> items;
        const uint32_t piece = lae::cdiv(nb, 8u);
        for (uint32_t l = 0; l < L; l++) {
            if (!dense[l]) items.push_back(Item{cost[l] * nb, l, false});
            else for (uint32_t k = 0; k < 8 && k * piece < nb; k++) items.push_back(Item{cost[l] * std::min(piece, nb - k * piece), l, true});
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
            else if (fs.nseg[best] < FWD_MAX_SEG) fs.seg[best][fs.nseg[best]++] = FwdSeg{it.level, 0u, nb};
            else ok = false;
        }
        for (uint32_t l = 0; l < L && ok; l++) {
            if (!dense[l]) continue;
            uint32_t c0 = 0;
            for (int x = 0; x < 8; x++) {
                if (!pieces[x][l]) continue;
                const uint32_t n = std::min(pieces[x][l] * piece, nb - c0);
                if (fs.nseg[x] < FWD_MAX_SEG) fs.seg[x][fs.nseg[x]++] = FwdSeg{l, c0, n}; else ok = false;
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
    const uint32_t nb = lae::cdiv(a.B_dev ? a.B_launch : a.B, GRID_BLOCK);
    if constexpr (std::is_same<T, half_t>::value && D == 3 && C == 2) {
        if (g_fwd_mode != 2 && !a.dy_dx && a.interp == 0 && !a.align && a.gridtype == 0 && a.L <= 8 * FWD_MAX_SEG && a.L <= MAX_LEVELS) {
            FwdSched fs;
            const uint32_t per_xcd = fwd_sched_build(fs, a.L, nb, a.sc, g_fwd_mode == 0 ? host_offsets(a.offsets, a.L, a.stream) : nullptr);
            k_grid_fwd_lean<<<per_xcd * 8, GRID_BLOCK, 0, a.stream>>>(a.inputs, (const half_t*)a.emb, a.offsets, (half_t*)a.out, a.B, a.sc,
                                                                      fs, a.os_b, a.os_l, a.B_dev);
            return;
        }
    }
    const bool xcd = (a.L % 8) == 0;
    k_grid_fwd<T, D, C><<<nb * a.L, GRID_BLOCK, 0, a.stream>>>(a.inputs, (const T*)a.emb, a.offsets, (T*)a.out, a.B, a.L,
                                                                a.sc, (T*)a.dy_dx, a.gridtype, a.align, a.interp, nb,
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
                        float in_scale = 1.0f) {
    if (B == 0) return LAE_OK;
    if (!inputs || !embeddings || !offsets || !outputs) return LAE_ENULL;
    if (gridtype > 1 || interp > 1) return LAE_EINVAL;
    FwdArgs a;
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
        lbc = lae::workspace(lae::WS_GRID_OUT_T, (size_t)B * L * C * (dtype == LAE_F16 ? 2 : 4));
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

// Level sizes on the host: offsets live in device memory (the reference passes a tensor), so the first call with a
// given (pointer, L) copies the L+1 ints once, synchronously, and later calls reuse them.  First calls happen during
// eager warm-up (workspaces are allocated there too), never inside a stream capture.
struct HostOffsets { const int32_t* ptr; uint32_t L; std::vector<int32_t> v; };
static const std::vector<int32_t>* host_offsets(const int32_t* offsets, uint32_t L, hipStream_t stream) {
    static std::vector<HostOffsets> cache;
    for (const auto& c : cache)
        if (c.ptr == offsets && c.L == L) return &c.v;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    HostOffsets h{offsets, L, std::vector<int32_t>(L + 1)};
    if (hipMemcpy(h.v.data(), offsets, sizeof(int32_t) * (L + 1), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;                                    // unknown: callers take the conservative path
    }
    if (cache.size() > 64) cache.clear();
    cache.push_back(std::move(h));
    return &cache.back().v;
}

// The binned backward in two halves.  PLAN (count pass + scans) reads the sample positions only and fills
// {bucket totals | bucket offsets | per-block exclusive counts}; EXEC (fill + accumulate) needs the gradients.  A caller
// may run PLAN early -- e.g. right after the march, beside the forward pass (lae_grid_encode_backward_plan) -- the default
// entry points run both back to back in the library workspace.
static inline size_t bin_tab_bytes(uint32_t L) { return (((size_t)2 * L * BK_MAX * 4 + 255) / 256) * 256 + 256; }   // + {ticket, done} of k_bin_acc
static inline size_t bin_plan_bytes(uint32_t B, uint32_t L) {
    const uint32_t nb = lae::cdiv(B, BIN_THREADS * BIN_SPT);
    return bin_tab_bytes(L) + (((size_t)nb * L * BK_MAX * 4 + 255) / 256) * 256;
}
struct BinPlan { uint32_t* counts; uint32_t* offs; uint32_t* block_counts; uint32_t* tickets; };
static inline BinPlan bin_plan_at(void* buf, uint32_t L) {
    uint8_t* p = reinterpret_cast<uint8_t*>(buf);
    BinPlan bp;
    bp.counts = reinterpret_cast<uint32_t*>(p);
    bp.offs = bp.counts + (size_t)L * BK_MAX;
    bp.block_counts = reinterpret_cast<uint32_t*>(p + bin_tab_bytes(L));
    bp.tickets = reinterpret_cast<uint32_t*>(p + bin_tab_bytes(L) - 256);
    return bp;
}

template <typename T>
static void bwd_plan(const float* inputs, const int32_t* offsets, uint32_t B, uint32_t L, const BwdArgs& a, const BinPlan& bp) {
    const uint32_t nb = lae::cdiv(B, BIN_THREADS * BIN_SPT);
    const size_t n_tab = (size_t)L * BK_MAX;
    k_bin<T, false><<<nb * L, BIN_THREADS, 0, a.stream>>>(nullptr, inputs, offsets, B, L, a.sc, a.gridtype, a.align, a.interp, nb,
                                                        bp.block_counts, bp.offs, nullptr);
    k_bin_scan_blocks<<<lae::cdiv(n_tab, 4), 256, 0, a.stream>>>(bp.block_counts, bp.counts, L, nb);
    k_bin_scan<<<1, 1024, 0, a.stream>>>(bp.counts, bp.offs, (uint32_t)n_tab, bp.tickets);
}

template <typename T>
static int bwd_exec(const void* gT, const float* inputs, const int32_t* offsets, void* gemb, uint32_t B, uint32_t L, const BwdArgs& a,
                    const BinPlan& bp, HItem<T>* queue) {
    const T* g = (const T*)gT;
    T* ge = (T*)gemb;
    const uint32_t nb = lae::cdiv(B, BIN_THREADS * BIN_SPT);
    k_bin<T, true><<<nb * L, BIN_THREADS, 0, a.stream>>>(g, inputs, offsets, B, L, a.sc, a.gridtype, a.align, a.interp, nb,
                                                       bp.block_counts, bp.offs, queue);
    k_bin_acc<T><<<(uint32_t)lae::num_cus(), LB_THREADS, 0, a.stream>>>(offsets, ge, L, a.sc, a.gridtype, a.align, bp.counts, bp.offs, queue,
                                                                        bp.tickets);
    // levels with more buckets than the tables hold (fp16: T > 2^21): generic atomic kernel; skipped when the host copy of
    // the level sizes shows that no level needs it
    bool need_generic = true;
    if (const std::vector<int32_t>* ho = host_offsets(offsets, L, a.stream)) {
        need_generic = false;
        for (uint32_t l = 0; l < L; l++) {
            const uint32_t size = (uint32_t)((*ho)[l + 1] - (*ho)[l]);
            const uint32_t res = (uint32_t)ceilf(a.sc.scale[l]) + 1;
            const uint64_t side = a.align ? res : res + 1;
            const bool use_hash = a.gridtype == 0 && side * side * side > size;
            if (!level_is_binned<T>(size, res, use_hash)) need_generic = true;
        }
    }
    if (need_generic) {
        const uint32_t nbg = lae::cdiv(B, GRID_BLOCK);
        k_grid_bwd<T, 3, 2><<<nbg * L, GRID_BLOCK, 0, a.stream>>>(g, inputs, offsets, ge, B, L, a.sc, a.gridtype, a.align, a.interp, nbg,
                                                                  (L % 8) == 0, 2, (uint64_t)B * 2, true);
    }
    return LAE_OK;
}

// queue of the fill / accumulate passes: worst case 8 items per (sample, level); library workspace
template <typename T>
static HItem<T>* bin_queue(uint32_t B, uint32_t L, size_t extra_bytes, uint8_t** base_out) {
    const size_t q_bytes = (size_t)B * 8 * L * sizeof(HItem<T>);
    uint8_t* ws = reinterpret_cast<uint8_t*>(lae::workspace(lae::WS_GRID_BINS, extra_bytes + q_bytes));
    if (base_out) *base_out = ws;
    return ws ? reinterpret_cast<HItem<T>*>(ws + extra_bytes) : nullptr;
}

template <typename T>
static int launch_bwd_fast(const void* gT, const float* inputs, const int32_t* offsets, void* gemb, uint32_t B, uint32_t L,
                           const BwdArgs& a, const void* plan = nullptr) {
    if (plan) {                                            // counts / offsets were prepared by lae_grid_encode_backward_plan
        HItem<T>* queue = bin_queue<T>(B, L, 0, nullptr);
        if (!queue) return LAE_ELAUNCH;
        return bwd_exec<T>(gT, inputs, offsets, gemb, B, L, a, bin_plan_at(const_cast<void*>(plan), L), queue);
    }
    uint8_t* ws = nullptr;
    HItem<T>* queue = bin_queue<T>(B, L, bin_plan_bytes(B, L), &ws);
    if (!queue) return LAE_ELAUNCH;
    const BinPlan bp = bin_plan_at(ws, L);
    bwd_plan<T>(inputs, offsets, B, L, a, bp);
    return bwd_exec<T>(gT, inputs, offsets, gemb, B, L, a, bp, queue);
}

// 0 = binned LDS pipeline where available (default), 1 = always the generic global-atomic kernel
static int g_force_atomic_bwd = 0;

static int grid_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                         void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                         const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                         int dtype, bool blc, void* stream, float in_shift = 0.0f, float in_scale = 1.0f, const void* plan = nullptr) {
    (void)embeddings;
    if (B == 0) return LAE_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings) return LAE_ENULL;
    if (plan && !(D == 3 && C == 2 && L <= 32 && !blc && !g_force_atomic_bwd)) return LAE_EINVAL;
    if (gridtype > 1 || interp > 1) return LAE_EINVAL;
    BwdArgs a;
    a.grad = grad; a.inputs = inputs; a.offsets = offsets; a.gemb = grad_embeddings; a.B = B; a.L = L;
    int rc = fill_scales(a.sc, L, S, H);
    if (rc) return rc;
    a.sc.in_shift = in_shift; a.sc.in_scale = in_scale;
    a.gridtype = gridtype; a.align = align_corners != 0; a.interp = interp;
    a.gs_b = blc ? (uint64_t)L * C : C;
    a.gs_l = blc ? C : (uint64_t)B * C;
    a.stream = reinterpret_cast<hipStream_t>(stream);
    if (dtype != LAE_F32 && dtype != LAE_F16) return LAE_EINVAL;
    if (D == 3 && C == 2 && L <= 32 && !g_force_atomic_bwd) {
        // LDS-partitioned path (see k_grid_bwd_lds); [B, L*2] grads are transposed into the workspace first
        const size_t esz = dtype == LAE_F16 ? 2 : 4;
        const void* gT = grad;
        if (blc) {
            void* ws = lae::workspace(lae::WS_GRID_GRAD_T, (size_t)B * L * 2 * esz);
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

// frame loop (raymarching.hip lae_render_frame): fp16 table, D=3, C=2, level-major output [L, B_cap, 2]; rows beyond
// *B_dev are not touched.  B_launch = host upper bound of *B_dev (sizes the launch only).
int lae::grid_forward_frame(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B_cap,
                            uint32_t B_launch, const uint32_t* B_dev, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                            int align_corners, uint32_t interp, float in_shift, float in_scale, hipStream_t stream) {
    if (B_launch == 0) return LAE_OK;
    FwdArgs a;
    a.inputs = inputs; a.emb = embeddings; a.offsets = offsets; a.out = outputs; a.B = B_cap; a.L = L;
    int rc = fill_scales(a.sc, L, S, H);
    if (rc) return rc;
    a.sc.in_shift = in_shift; a.sc.in_scale = in_scale;
    a.dy_dx = nullptr; a.gridtype = gridtype; a.align = align_corners != 0; a.interp = interp;
    a.stream = stream; a.os_b = 2; a.os_l = (uint64_t)B_cap * 2;
    a.B_dev = B_dev; a.B_launch = std::min(B_launch, B_cap);
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
int lae_grid_encode_forward_blc(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float in_shift,
                                float in_scale, void* stream) {
    return grid_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp,
                        dtype, true, stream, in_shift, in_scale);
}
int lae_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                             void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                             uint32_t interp, int dtype, void* stream) {
    return grid_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp, dtype, false, stream);
}
int lae_grid_encode_backward_blc(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                 void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                 uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                                 int align_corners, uint32_t interp, int dtype, float in_shift, float in_scale,
                                 void* stream) {
    return grid_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp, dtype, true, stream, in_shift, in_scale);
}

int lae_grid_encode_forward_ex(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                               uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int blc, float in_shift,
                               float in_scale, void* stream) {
    return grid_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp,
                        dtype, blc != 0, stream, in_shift, in_scale);
}
int lae_grid_encode_backward_ex(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                                int align_corners, uint32_t interp, int dtype, int blc, float in_shift, float in_scale,
                                void* stream) {
    return grid_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp, dtype, blc != 0, stream, in_shift, in_scale);
}

uint64_t lae_grid_backward_plan_bytes(uint32_t B, uint32_t L) { return bin_plan_bytes(B, L); }

int lae_grid_encode_backward_plan(const float* inputs, const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                  uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, float in_shift,
                                  float in_scale, void* plan, void* stream) {
    if (B == 0) return LAE_OK;
    if (!inputs || !offsets || !plan) return LAE_ENULL;
    if (D != 3 || C != 2 || L > 32 || gridtype > 1 || interp > 1 || (dtype != LAE_F32 && dtype != LAE_F16)) return LAE_EINVAL;
    BwdArgs a;
    a.grad = nullptr; a.inputs = inputs; a.offsets = offsets; a.gemb = nullptr; a.B = B; a.L = L;
    int rc = fill_scales(a.sc, L, S, H);
    if (rc) return rc;
    a.sc.in_shift = in_shift; a.sc.in_scale = in_scale;
    a.gridtype = gridtype; a.align = align_corners != 0; a.interp = interp;
    a.gs_b = C; a.gs_l = (uint64_t)B * C;
    a.stream = reinterpret_cast<hipStream_t>(stream);
    if (dtype == LAE_F16) bwd_plan<half_t>(inputs, offsets, B, L, a, bin_plan_at(plan, L));
    else bwd_plan<float>(inputs, offsets, B, L, a, bin_plan_at(plan, L));
    return lae::check_launch("grid_encode_backward_plan");
}

int lae_grid_encode_backward_planned(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                     uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                     uint32_t interp, int dtype, float in_shift, float in_scale, const void* plan, void* stream) {
    if (!plan) return LAE_ENULL;
    return grid_backward(grad, inputs, nullptr, offsets, grad_embeddings, B, D, C, L, S, H, nullptr, nullptr, gridtype, align_corners,
                         interp, dtype, false, stream, in_shift, in_scale, plan);
}

int lae_grid_forward_schedule(const int32_t* offsets_host, uint32_t L, float S, uint32_t H, uint32_t n_chunks, uint32_t* nseg_out,
                              uint32_t* segs_out) {
    if (!offsets_host || !nseg_out || !segs_out) return LAE_ENULL;
    LevelScales sc;
    if (fill_scales(sc, L, S, H) != LAE_OK || L > 8 * FWD_MAX_SEG) return LAE_EINVAL;
    const std::vector<int32_t> offs(offsets_host, offsets_host + L + 1);
    FwdSched fs;
    const uint32_t per_xcd = fwd_sched_build(fs, L, n_chunks, sc, &offs);
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
