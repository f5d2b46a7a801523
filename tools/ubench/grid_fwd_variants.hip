// microbenchmark: hash-grid forward (fp16 table, D=3, C=2, L=16, T=2^19) on ray-coherent samples -- level -> XCD maps,
// per-level solo cost and gather variants.  Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/gfv tools/ubench/grid_fwd_variants.hip && /tmp/gfv
// Every variant's output is compared bit for bit with variant 0 under map 0.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

constexpr int L = 16;
constexpr int BLOCK = 256;
struct Scales { float scale[L]; };
struct LevelMap { uint8_t n[8]; uint8_t lv[8][16]; uint32_t interleave; };     // levels handled by the blocks with blockIdx % 8 == x, in order

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t cell(bool hash, uint32_t stride1, uint32_t stride2, uint32_t size, uint32_t x, uint32_t y, uint32_t z) {
    uint32_t i = hash ? (x ^ (y * 2654435761u) ^ (z * 805459861u)) : (x + y * stride1 + z * stride2);
    return (size & (size - 1)) == 0 ? (i & (size - 1)) : (i % size);
}

// VAR bits: 1 = 16-byte load when both entries sit in one aligned group of four (3-way branch)
//           2 = chunk loop (`loops` chunks per block)
//           4 = coordinates as one 12-byte load
//           8 = dense levels: the pair is always adjacent -> one 4-byte-aligned 8-byte load
//          16 = (diagnostic, wrong results) only the x corner is loaded: 4 loads, 4 lines per lane
//          32 = (diagnostic) never pair: 8 loads per lane
//          64 = non-temporal gathers
//         128 = hashed levels: always the aligned 16-byte group of x, plus a 4-byte load only where x+1 leaves the group
struct __attribute__((packed, aligned(4))) U2 { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename V> __device__ __forceinline__ V ld(const V* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <int VAR>
__global__ __launch_bounds__(BLOCK) void k_fwd(const float* __restrict__ xyz, const half2_t* __restrict__ grid,
                                               const int32_t* __restrict__ offsets, half2_t* __restrict__ out, uint32_t B,
                                               Scales sc, LevelMap map, uint32_t nb, uint32_t loops) {
    constexpr bool NT = (VAR & 64) != 0;
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t nlv = map.n[xcd];
    if (nlv == 0) return;
    const uint32_t slot = map.interleave ? j % nlv : j / nb;       // interleave: consecutive blocks of an XCD alternate levels
    const uint32_t jc = map.interleave ? j / nlv : j % nb;
    if (slot >= nlv || jc >= nb) return;
    const uint32_t level = map.lv[xcd][slot];
    const float scale = sc.scale[level];
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off;
    const uint32_t s1 = res + 1, s2 = (res + 1) * (res + 1);
    const bool hash = (uint64_t)s2 * (res + 1) > size;
    const half2_t* __restrict__ tab = grid + off;
    const uint32_t* __restrict__ tabw = reinterpret_cast<const uint32_t*>(tab);
    for (uint32_t it = 0; it < loops; it++) {
        const uint32_t chunk = jc + it * nb;
        const uint32_t b = chunk * BLOCK + threadIdx.x;
        if (b >= B) return;
        float x[3]; uint32_t pg[3]; float fr[3];
        if (VAR & 4) { const F3 v = *reinterpret_cast<const F3*>(xyz + (size_t)b * 3); x[0] = v.x; x[1] = v.y; x[2] = v.z; }
        else { x[0] = xyz[(size_t)b * 3]; x[1] = xyz[(size_t)b * 3 + 1]; x[2] = xyz[(size_t)b * 3 + 2]; }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            x[d] = (x[d] + 1.0f) * 0.5f;
            float p = fmaf(x[d], scale, 0.5f);
            const float fl = floorf(p);
            pg[d] = (uint32_t)fl;
            fr[d] = p - fl;
        }
        uint32_t cw[8];
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            const uint32_t y = pg[1] + (yz & 1), z = pg[2] + (yz >> 1);
            const uint32_t i0 = cell(hash, s1, s2, size, pg[0], y, z), i1 = cell(hash, s1, s2, size, pg[0] + 1, y, z);
            uint32_t a, c;
            if (VAR & 16) { a = ld(tabw + i0, NT); c = a; }
            else if (VAR & 32) { a = ld(tabw + i0, NT); c = ld(tabw + i1, NT); }
            else if ((VAR & 8) && !hash) {
                const U2 w = *reinterpret_cast<const U2*>(tabw + i0);     // i1 == i0 + 1 on a dense level that never wraps
                a = w.x; c = w.y;
            } else if ((VAR & 128) && hash) {
                const u32x4 w = ld(reinterpret_cast<const u32x4*>(tabw + (i0 & ~3u)), NT);
                const uint32_t e0 = i0 & 3u, e1 = i1 & 3u;
                a = e0 == 0 ? w.x : e0 == 1 ? w.y : e0 == 2 ? w.z : w.w;
                c = e1 == 0 ? w.x : e1 == 1 ? w.y : e1 == 2 ? w.z : w.w;
                if ((i1 >> 2) != (i0 >> 2)) c = ld(tabw + i1, NT);
            } else if (i1 == (i0 ^ 1u)) {
                const u32x2 w = ld(reinterpret_cast<const u32x2*>(tabw + (i0 & ~1u)), NT);
                a = (i0 & 1u) ? w.y : w.x; c = (i0 & 1u) ? w.x : w.y;
            } else if ((VAR & 1) && (i1 >> 2) == (i0 >> 2)) {
                const u32x4 w = ld(reinterpret_cast<const u32x4*>(tabw + (i0 & ~3u)), NT);
                const uint32_t e0 = i0 & 3u, e1 = i1 & 3u;
                a = e0 == 0 ? w.x : e0 == 1 ? w.y : e0 == 2 ? w.z : w.w;
                c = e1 == 0 ? w.x : e1 == 1 ? w.y : e1 == 2 ? w.z : w.w;
            } else { a = ld(tabw + i0, NT); c = ld(tabw + i1, NT); }
            cw[2 * yz] = a; cw[2 * yz + 1] = c;
        }
        half_t r0 = (half_t)0.f, r1 = (half_t)0.f;
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const float w = (((idx & 1) ? fr[0] : 1 - fr[0]) * ((idx & 2) ? fr[1] : 1 - fr[1])) * ((idx & 4) ? fr[2] : 1 - fr[2]);
            const half2_t v = __builtin_bit_cast(half2_t, cw[idx]);
            r0 = r0 + (half_t)(w * (float)v[0]);              // Half += float: product rounded to half first (gridencoder.cu:187)
            r1 = r1 + (half_t)(w * (float)v[1]);
        }
        half2_t h = {r0, r1};
        out[(size_t)level * B + b] = h;
        if (!(VAR & 2)) return;
    }
}

// VAR 256: lean specialisation -- hashed power-of-two levels: y / z hash terms by one multiply + one add each, byte offsets
// formed before the xor (no shifts per corner), pairing decided once per lane (x even), 32-bit offsets against a uniform
// base; dense levels: eight plain loads (no divergent pair path), no modulo.
__global__ __launch_bounds__(BLOCK) void k_fwd_lean(const float* __restrict__ xyz, const half2_t* __restrict__ grid,
                                                    const int32_t* __restrict__ offsets, half2_t* __restrict__ out, uint32_t B,
                                                    Scales sc, LevelMap map, uint32_t nb, uint32_t loops) {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t nlv = map.n[xcd];
    if (nlv == 0) return;
    const uint32_t slot = map.interleave ? j % nlv : j / nb;
    const uint32_t jc = map.interleave ? j / nlv : j % nb;
    if (slot >= nlv || jc >= nb) return;
    const uint32_t level = map.lv[xcd][slot];
    const float scale = sc.scale[level];
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off;
    const uint32_t s1 = res + 1, s2 = (res + 1) * (res + 1);
    const bool hash = (uint64_t)s2 * (res + 1) > size;
    const char* __restrict__ tabb = reinterpret_cast<const char*>(grid + off);
    const uint32_t b = jc * BLOCK + threadIdx.x;
    if (b >= B) return;
    float fr[3]; uint32_t pg[3];
    const F3 v3 = *reinterpret_cast<const F3*>(xyz + (size_t)b * 3);
    const float xin[3] = {v3.x, v3.y, v3.z};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float x01 = (xin[d] + 1.0f) * 0.5f;
        const float p = fmaf(x01, scale, 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)fl;
        fr[d] = p - fl;
    }
    uint32_t cw[8];
    if (hash) {                                            // size is a power of two on every hashed level of this table
        const uint32_t m4 = (size - 1u) << 2;
        const uint32_t hy0 = (pg[1] * 2654435761u) << 2, hy1 = hy0 + (2654435761u << 2);
        const uint32_t hz0 = (pg[2] * 805459861u) << 2, hz1 = hz0 + (805459861u << 2);
        const uint32_t x0 = pg[0] << 2, x1 = x0 + 4u;
        const uint32_t h[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        if ((pg[0] & 1u) == 0) {
#pragma unroll
            for (int yz = 0; yz < 4; yz++) {
                const uint32_t o0 = (x0 ^ h[yz]) & m4;     // entry of x; x + 1 sits in the other half of the aligned pair
                const u32x2 w = *reinterpret_cast<const u32x2*>(tabb + (o0 & ~4u));
                cw[2 * yz] = (o0 & 4u) ? w.y : w.x; cw[2 * yz + 1] = (o0 & 4u) ? w.x : w.y;
            }
        } else {
#pragma unroll
            for (int yz = 0; yz < 4; yz++) {
                cw[2 * yz] = *reinterpret_cast<const uint32_t*>(tabb + ((x0 ^ h[yz]) & m4));
                cw[2 * yz + 1] = *reinterpret_cast<const uint32_t*>(tabb + ((x1 ^ h[yz]) & m4));
            }
        }
    } else {
        const uint32_t base = (pg[0] + pg[1] * s1 + pg[2] * s2) << 2;
        const uint32_t dy = s1 << 2, dz = s2 << 2;
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            const uint32_t o0 = base + ((yz & 1) ? dy : 0u) + ((yz >> 1) ? dz : 0u);
            cw[2 * yz] = *reinterpret_cast<const uint32_t*>(tabb + o0);
            cw[2 * yz + 1] = *reinterpret_cast<const uint32_t*>(tabb + o0 + 4u);
        }
    }
    half_t r0 = (half_t)0.f, r1 = (half_t)0.f;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        const float w = (((idx & 1) ? fr[0] : 1 - fr[0]) * ((idx & 2) ? fr[1] : 1 - fr[1])) * ((idx & 4) ? fr[2] : 1 - fr[2]);
        const half2_t v = __builtin_bit_cast(half2_t, cw[idx]);
        r0 = r0 + (half_t)(w * (float)v[0]);
        r1 = r1 + (half_t)(w * (float)v[1]);
    }
    half2_t h2 = {r0, r1};
    out[(size_t)level * B + b] = h2;
}

// VAR 512 (round 3): the north star's "LDS-staged per-level features", measured instead of priced.  The tables
// of the two coarsest levels (4920 + 13824 entries = 75 KB as half2) are copied into LDS by every workgroup (coalesced 16-byte
// loads), then the workgroup walks its share of the samples and gathers the 2 x 8 corners with ds_read2_b32 (x and x + 1 are
// adjacent on a dense level: one LDS instruction per corner row) -- same coordinate arithmetic, same accumulation order, so the
// results are the same bits as the lean kernel's.  Persistent grid: `blocks` workgroups (1 or 2 per CU: 75 KB each).
constexpr int LDS_LEVELS = 2;
__global__ __launch_bounds__(BLOCK) void k_fwd_lds(const float* __restrict__ xyz, const half2_t* __restrict__ grid,
                                                   const int32_t* __restrict__ offsets, half2_t* __restrict__ out, uint32_t B, Scales sc) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];
    const uint32_t n_ent = (uint32_t)offsets[LDS_LEVELS];                       // levels 0 .. LDS_LEVELS-1 are contiguous from entry 0
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(grid);
        u32x4* dst = reinterpret_cast<u32x4*>(tab);
        for (uint32_t i = threadIdx.x; i < n_ent / 4; i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    uint32_t s1[LDS_LEVELS], s2[LDS_LEVELS], off[LDS_LEVELS]; float scale[LDS_LEVELS];
#pragma unroll
    for (int l = 0; l < LDS_LEVELS; l++) {
        scale[l] = sc.scale[l];
        const uint32_t res = (uint32_t)ceilf(scale[l]) + 1;
        s1[l] = res + 1; s2[l] = (res + 1) * (res + 1); off[l] = (uint32_t)offsets[l];
    }
    for (uint32_t b = blockIdx.x * BLOCK + threadIdx.x; b < B; b += gridDim.x * BLOCK) {
        const F3 v3 = *reinterpret_cast<const F3*>(xyz + (size_t)b * 3);
        const float xin[3] = {v3.x, v3.y, v3.z};
#pragma unroll
        for (int l = 0; l < LDS_LEVELS; l++) {
            float fr[3]; uint32_t pg[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float x01 = (xin[d] + 1.0f) * 0.5f;
                const float p = fmaf(x01, scale[l], 0.5f);
                const float fl = floorf(p);
                pg[d] = (uint32_t)fl;
                fr[d] = p - fl;
            }
            const uint32_t base = off[l] + pg[0] + pg[1] * s1[l] + pg[2] * s2[l];
            uint32_t cw[8];
#pragma unroll
            for (int yz = 0; yz < 4; yz++) {
                const uint32_t o0 = base + ((yz & 1) ? s1[l] : 0u) + ((yz >> 1) ? s2[l] : 0u);
                cw[2 * yz] = tab[o0]; cw[2 * yz + 1] = tab[o0 + 1];          // adjacent dwords: the compiler emits ds_read2_b32
            }
            half_t r0 = (half_t)0.f, r1 = (half_t)0.f;
#pragma unroll
            for (int idx = 0; idx < 8; idx++) {
                const float w = (((idx & 1) ? fr[0] : 1 - fr[0]) * ((idx & 2) ? fr[1] : 1 - fr[1])) * ((idx & 4) ? fr[2] : 1 - fr[2]);
                const half2_t v = __builtin_bit_cast(half2_t, cw[idx]);
                r0 = r0 + (half_t)(w * (float)v[0]);
                r1 = r1 + (half_t)(w * (float)v[1]);
            }
            const half2_t h2 = {r0, r1};
            out[(size_t)l * B + b] = h2;
        }
    }
}

static LevelMap map_levels(const std::vector<std::vector<int>>& per_xcd, uint32_t interleave = 0) {
    LevelMap m; memset(&m, 0, sizeof(m)); m.interleave = interleave;
    for (int x = 0; x < 8; x++) { m.n[x] = (uint8_t)per_xcd[x].size(); for (size_t i = 0; i < per_xcd[x].size(); i++) m.lv[x][i] = (uint8_t)per_xcd[x][i]; }
    return m;
}
static uint32_t max_slots(const LevelMap& m) { uint32_t s = 0; for (int x = 0; x < 8; x++) s = m.n[x] > s ? m.n[x] : s; return s; }

int main(int argc, char** argv) {
    const uint32_t T = 1u << (argc > 2 ? atoi(argv[2]) : 19), H = 16;
    const float pls = exp2f(log2f(2048.0f / 16.0f) / 15.0f), S = log2f(pls);
    Scales sc; std::vector<int32_t> offs(L + 1); int32_t o = 0;
    for (int l = 0; l < L; l++) {
        sc.scale[l] = fmaf(exp2f((float)l * S), (float)H, -1.0f);
        const uint32_t res = (uint32_t)ceil((double)H * pow((double)pls, l));
        uint64_t n = (uint64_t)(res + 1) * (res + 1) * (res + 1); if (n > T) n = T;
        n = (n + 7) / 8 * 8;
        offs[l] = o; o += (int32_t)n;
    }
    offs[L] = o;
    // ray-coherent samples: camera on a sphere of radius 3.2, constant step, samples kept inside a ball of radius 0.6
    const uint32_t NR = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
    const float dt = 2.0f * sqrtf(3.0f) / 1024.0f;
    std::vector<float> xyz; uint32_t rng = 12345;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 16777216.0f; };
    auto rnd_dir = [&](float* v) { for (;;) { v[0] = 2 * rnd() - 1; v[1] = 2 * rnd() - 1; v[2] = 2 * rnd() - 1; const float n = v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; if (n > 1e-3f && n <= 1.f) { const float s = 1.f / sqrtf(n); v[0] *= s; v[1] *= s; v[2] *= s; return; } } };
    float cam[3]; rnd_dir(cam); for (int d = 0; d < 3; d++) cam[d] *= 3.2f;     // one view per batch, like the loader
    for (uint32_t r = 0; r < NR; r++) {
        float tg[3]; rnd_dir(tg); const float rad = 1.1f * cbrtf(rnd());
        float dir[3]; float n = 0;
        for (int d = 0; d < 3; d++) { dir[d] = tg[d] * rad - cam[d]; n += dir[d] * dir[d]; }
        n = 1.f / sqrtf(n); for (int d = 0; d < 3; d++) dir[d] *= n;
        const float t0 = 2.0f + dt * rnd();
        for (int k = 0; k < 1024; k++) {
            const float t = t0 + k * dt; float p[3]; float rr = 0;
            for (int d = 0; d < 3; d++) { p[d] = cam[d] + t * dir[d]; rr += p[d] * p[d]; }
            if (rr < 0.36f) { xyz.push_back(p[0]); xyz.push_back(p[1]); xyz.push_back(p[2]); }
        }
    }
    const uint32_t B = (uint32_t)(xyz.size() / 3);
    printf("%u rays -> %u samples (%.1f per ray); table %d entries (%.1f MB fp16)\n", NR, B, (double)B / NR, o, o * 4 / 1e6);
    std::vector<uint16_t> tabh((size_t)o * 2);
    for (auto& v : tabh) { rng = rng * 1664525u + 1013904223u; v = (uint16_t)(0x2000u + ((rng >> 12) & 0x0fffu)) | (uint16_t)((rng >> 31) << 15); }
    float* d_xyz; half2_t* d_tab; int32_t* d_off; half2_t *d_out, *d_ref;
    CK(hipMalloc(&d_xyz, xyz.size() * 4)); CK(hipMalloc(&d_tab, (size_t)o * 4)); CK(hipMalloc(&d_off, (L + 1) * 4));
    CK(hipMalloc(&d_out, (size_t)L * B * 4)); CK(hipMalloc(&d_ref, (size_t)L * B * 4));
    CK(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, tabh.data(), (size_t)o * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, offs.data(), (L + 1) * 4, hipMemcpyHostToDevice));
    const uint32_t nchunks = (B + BLOCK - 1) / BLOCK;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint32_t> href((size_t)L * B), hout((size_t)L * B);

    auto run = [&](int var, const LevelMap& m, uint32_t loops, half2_t* dst, const char* name, uint32_t nlevels_alg) {
        const uint32_t nb = (nchunks + loops - 1) / loops;
        const uint32_t blocks = 8 * nb * max_slots(m);
        auto launch = [&]() {
#define CASE(V) case V: k_fwd<V><<<blocks, BLOCK>>>(d_xyz, d_tab, d_off, dst, B, sc, m, nb, loops); break;
            switch (var) {
                CASE(0) CASE(1) CASE(2) CASE(4) CASE(8) CASE(12) CASE(16) CASE(32) CASE(64) CASE(128) CASE(140) CASE(142) CASE(204)
                case 256: k_fwd_lean<<<blocks, BLOCK>>>(d_xyz, d_tab, d_off, dst, B, sc, m, nb, loops); break;
                default: printf("variant %d not instantiated\n", var); exit(1);
            }
        };
        float best = 1e30f, sum = 0;
        for (int rep = 0; rep < 3; rep++) {
            for (int i = 0; i < 5; i++) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 40; i++) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float us = ms * 1000.f / 40.f; best = us < best ? us : best; sum += us;
        }
        CK(hipGetLastError());
        printf("%-46s var %d loops %2u: %7.1f us (best of 3, mean %.1f)  %6.0f GB/s alg\n", name, var, loops, best, sum / 3,
               (double)B * nlevels_alg * (12.0 / 16 + 36.0) / best / 1e3);
        return best;
    };
    // map 0: production (level l and l+8 on XCD l)
    std::vector<std::vector<int>> prod(8); for (int x = 0; x < 8; x++) prod[x] = {x, x + 8};
    const LevelMap m_prod = map_levels(prod);
    CK(hipMemset(d_ref, 0, (size_t)L * B * 4));
    run(0, m_prod, 1, d_ref, "map prod (l, l+8)", 16);
    CK(hipMemcpy(href.data(), d_ref, (size_t)L * B * 4, hipMemcpyDeviceToHost));
    auto check = [&](const char* what) {
        CK(hipMemcpy(hout.data(), d_out, (size_t)L * B * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < hout.size(); i++) bad += hout[i] != href[i];
        if (bad) printf("    MISMATCH in %s: %zu of %zu words differ\n", what, bad, hout.size());
    };
    auto full = [&](int var, const LevelMap& m, uint32_t loops, const char* name) {
        CK(hipMemset(d_out, 0, (size_t)L * B * 4));
        run(var, m, loops, d_out, name, 16);
        check(name);
    };
    full(256, m_prod, 1, "prod map, LEAN kernel");
    { std::vector<std::vector<int>> v(8); for (int x = 0; x < 8; x++) v[x] = {x + 8, x}; full(0, map_levels(v), 1, "(l+8, l) fine first");
      full(256, map_levels(v), 1, "(l+8, l) fine first, LEAN"); }
    {   // ---- LDS staging of levels 0 + 1 (VAR 512) against the lean kernel on the same two levels
        // lean kernel, levels 0 and 1 on EVERY XCD (8x the work of the two levels): time / 8 = their cost spread over the chip,
        // which is how the production schedule deals them (eighths of the dense levels fill the XCDs' gaps)
        std::vector<std::vector<int>> v(8); for (int x = 0; x < 8; x++) v[x] = {0, 1};
        const float t8 = run(256, map_levels(v), 1, d_out, "LEAN, levels 0+1 on every XCD (8x work)", 16);
        printf("    -> levels 0+1 spread over the chip: %.2f us\n", t8 / 8);
        hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
        const size_t lds_bytes = (size_t)offs[LDS_LEVELS] * 4;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        for (int per_cu : {1, 2}) {
            const uint32_t blocks = (uint32_t)prop.multiProcessorCount * per_cu;
            CK(hipMemset(d_out, 0, (size_t)L * B * 4));
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                for (int i = 0; i < 5; i++) k_fwd_lds<<<blocks, BLOCK, lds_bytes>>>(d_xyz, d_tab, d_off, d_out, B, sc);
                CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
                for (int i = 0; i < 40; i++) k_fwd_lds<<<blocks, BLOCK, lds_bytes>>>(d_xyz, d_tab, d_off, d_out, B, sc);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms * 25.f < best ? ms * 25.f : best;
            }
            CK(hipGetLastError());
            CK(hipMemcpy(hout.data(), d_out, (size_t)L * B * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < (size_t)LDS_LEVELS * B; i++) bad += hout[i] != href[i];
            printf("LDS-staged levels 0+1 (%zu KB per workgroup), %u workgroups (%d per CU): %7.1f us  %s\n", lds_bytes / 1024, blocks, per_cu, best,
                   bad ? "MISMATCH" : "same bits as the reference kernel");
        }
        // the same kernel without any sample (staging only): what the 75 KB copy per workgroup costs
        for (int per_cu : {1, 2}) {
            const uint32_t blocks = (uint32_t)prop.multiProcessorCount * per_cu;
            for (int i = 0; i < 5; i++) k_fwd_lds<<<blocks, BLOCK, lds_bytes>>>(d_xyz, d_tab, d_off, d_out, 0, sc);
            CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
            for (int i = 0; i < 40; i++) k_fwd_lds<<<blocks, BLOCK, lds_bytes>>>(d_xyz, d_tab, d_off, d_out, 0, sc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("    staging alone (no samples), %d per CU: %7.1f us\n", per_cu, ms * 25.f);
        }
    }
    for (int var : {0, 256}) {
        for (int l : {0, 3, 5, 8, 11, 15}) {
            std::vector<std::vector<int>> v(8); v[l & 7] = {l};
            char name[64]; snprintf(name, sizeof name, "solo level %d", l);
            run(var, map_levels(v), 1, d_out, name, 1);
        }
    }
    return 0;
    // solo levels and subsets (time only)
    for (int l = 0; l < 16; l++) {
        std::vector<std::vector<int>> v(8); v[l & 7] = {l};
        char name[64]; snprintf(name, sizeof name, "solo level %d on one XCD", l);
        run(0, map_levels(v), 1, d_out, name, 1);
    }
    { std::vector<std::vector<int>> v(8); for (int x = 2; x < 8; x++) v[x] = {x + 8}; run(0, map_levels(v), 1, d_out, "levels 10-15, one per XCD", 6); }
    { std::vector<std::vector<int>> v(8); for (int x = 0; x < 8; x++) v[x] = {x + 8}; run(0, map_levels(v), 1, d_out, "levels 8-15, one per XCD", 8); }
    { std::vector<std::vector<int>> v(8); for (int x = 0; x < 8; x++) v[x] = {x}; run(0, map_levels(v), 1, d_out, "levels 0-7, one per XCD", 8); }
    { std::vector<std::vector<int>> v(8); v[7] = {7, 15}; run(0, map_levels(v), 1, d_out, "levels 7+15 on one XCD", 2); }
    { std::vector<std::vector<int>> v(8); for (int x = 0; x < 8; x++) v[x] = {15}; run(0, map_levels(v), 1, d_out, "level 15 on every XCD (8x the work)", 8); }
    return 0;
}
