// raymarch_common.h -- device-side pieces shared by raymarching.hip (the operators) and frame.hip (the device-resident
// inference loop): Morton codes, the cascaded-occupancy probe, the marcher geometry.  Included INSIDE each file's anonymous
// namespace user; every translation unit gets its own copy (all __forceinline__ / static).
#pragma once
#include <climits>
#include "lae_common.h"

namespace {

using lae::clampf;

__device__ __forceinline__ uint32_t spread10(uint32_t v) {   // raymarching.cu:56-63
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread10(x) | (spread10(y) << 1) | (spread10(z) << 2);
}
__device__ __forceinline__ uint32_t morton_compact(uint32_t x) {  // raymarching.cu:73-81
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}


// ---------------------------------------------------------------- the marcher
// One probe of the cascaded occupancy bitfield at ray parameter t
// (raymarching.cu:361-399); see oracle/lae_oracle.c marcher_probe for the
// arithmetic contract.
struct Ray {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};
struct MarchCfg {
    float bound, dt_gamma, dt_min, dt_max, rH, Hf, Cf, Hm1;
    uint32_t H3;
    int bound_exp;       // bound == 2^bound_exp, or INT_MIN when bound is not a power of two (mip_bounds then divides)
};
struct Probe {
    float x, y, z, dt, tt;
    uint32_t index;
    bool occ;
};

__device__ __forceinline__ int cascade_of(float v, float Cf) {
    int e;
    (void)frexpf(v, &e);
    return (int)fminf(Cf - 1.0f, fmaxf(0.0f, (float)e));
}

// mip_bound = min(2^level, bound) and its reciprocal (raymarching.cu:370-371: `1 / mip_bound`).  With bound a power of two (every
// shipped config: 1 or 2) both are powers of two: 2^min(level, bound_exp) and 2^-min(level, bound_exp) -- the same bits as the
// IEEE division, one v_ldexp_f32 instead of its ten dependent instructions in every visit of every walk.
__device__ __forceinline__ void mip_bounds(const MarchCfg& c, int level, float& mip_bound, float& mip_rbound) {
    if (c.bound_exp != INT_MIN) {
        const int l = level < c.bound_exp ? level : c.bound_exp;
        mip_bound = scalbnf(1.0f, l);
        mip_rbound = scalbnf(1.0f, -l);
    } else {
        mip_bound = fminf(scalbnf(1.0f, level), c.bound);
        mip_rbound = 1.0f / mip_bound;
    }
}

__device__ __forceinline__ Probe probe_at(const Ray& r, const MarchCfg& c, const uint8_t* __restrict__ grid, float t) {
    Probe p;
    p.x = clampf(fmaf(t, r.dx, r.ox), -c.bound, c.bound);
    p.y = clampf(fmaf(t, r.dy, r.oy), -c.bound, c.bound);
    p.z = clampf(fmaf(t, r.dz, r.oz), -c.bound, c.bound);
    p.dt = clampf(t * c.dt_gamma, c.dt_min, c.dt_max);
    const float amax = fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z)));
    const int lp = cascade_of(amax, c.Cf);
    const int ld = cascade_of(p.dt * c.Hf * 0.5f, c.Cf);
    const int level = lp > ld ? lp : ld;
    float mip_bound, mip_rbound;
    mip_bounds(c, level, mip_bound, mip_rbound);
    const int nx = (int)clampf((0.5f * fmaf(p.x, mip_rbound, 1.0f)) * c.Hf, 0.0f, c.Hm1);
    const int ny = (int)clampf((0.5f * fmaf(p.y, mip_rbound, 1.0f)) * c.Hf, 0.0f, c.Hm1);
    const int nz = (int)clampf((0.5f * fmaf(p.z, mip_rbound, 1.0f)) * c.Hf, 0.0f, c.Hm1);
    p.index = (uint32_t)level * c.H3 + morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    p.occ = (grid[p.index >> 3] >> (p.index & 7u)) & 1u;
    if (!p.occ) {
        const float ax = (float)nx + 0.5f + 0.5f * copysignf(1.0f, r.dx);
        const float ay = (float)ny + 0.5f + 0.5f * copysignf(1.0f, r.dy);
        const float az = (float)nz + 0.5f + 0.5f * copysignf(1.0f, r.dz);
        const float tx = fmaf((ax * c.rH) * 2 - 1, mip_bound, -p.x) * r.rdx;
        const float ty = fmaf((ay * c.rH) * 2 - 1, mip_bound, -p.y) * r.rdy;
        const float tz = fmaf((az * c.rH) * 2 - 1, mip_bound, -p.z) * r.rdz;
        p.tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    } else {
        p.tt = t;
    }
    return p;
}

// the geometry of a visit without its occupancy probe: bit index of the cell, the step there, and where the walker lands if the
// cell turns out EMPTY (probe_at's arithmetic, statement for statement).  None of it depends on the bitfield, so a lane can lay
// out the next few visits of a walk through empty space and have all their probes in flight at once (k_frame_lookahead).
struct VisitGeom { uint32_t index; float dt, tt_empty; };
__device__ __forceinline__ VisitGeom visit_geom(const Ray& r, const MarchCfg& c, float t) {
    VisitGeom v;
    const float x = clampf(fmaf(t, r.dx, r.ox), -c.bound, c.bound);
    const float y = clampf(fmaf(t, r.dy, r.oy), -c.bound, c.bound);
    const float z = clampf(fmaf(t, r.dz, r.oz), -c.bound, c.bound);
    v.dt = clampf(t * c.dt_gamma, c.dt_min, c.dt_max);
    const float amax = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    const int lp = cascade_of(amax, c.Cf);
    const int ld = cascade_of(v.dt * c.Hf * 0.5f, c.Cf);
    const int level = lp > ld ? lp : ld;
    float mip_bound, mip_rbound;
    mip_bounds(c, level, mip_bound, mip_rbound);
    const int nx = (int)clampf((0.5f * fmaf(x, mip_rbound, 1.0f)) * c.Hf, 0.0f, c.Hm1);
    const int ny = (int)clampf((0.5f * fmaf(y, mip_rbound, 1.0f)) * c.Hf, 0.0f, c.Hm1);
    const int nz = (int)clampf((0.5f * fmaf(z, mip_rbound, 1.0f)) * c.Hf, 0.0f, c.Hm1);
    v.index = (uint32_t)level * c.H3 + morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const float ax = (float)nx + 0.5f + 0.5f * copysignf(1.0f, r.dx);
    const float ay = (float)ny + 0.5f + 0.5f * copysignf(1.0f, r.dy);
    const float az = (float)nz + 0.5f + 0.5f * copysignf(1.0f, r.dz);
    const float tx = fmaf((ax * c.rH) * 2 - 1, mip_bound, -x) * r.rdx;
    const float ty = fmaf((ay * c.rH) * 2 - 1, mip_bound, -y) * r.rdy;
    const float tz = fmaf((az * c.rH) * 2 - 1, mip_bound, -z) * r.rdz;
    v.tt_empty = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    return v;
}

// `do t += dt while (t < tt)` (raymarching.cu:396-398).  Constant step (dt_gamma == 0, every shipped config): inside one binade the
// walk is T_k = t + k q exactly (uniform_step below), so the first T_k >= tt (k >= 1) follows from one multiply and at most one
// correction either way instead of 5-9 dependent add / compare / branch rounds per empty cell -- the same value bit for bit, or
// the loop itself whenever the jump could leave the binade, dt sits on a rounding tie, or k is large.
__device__ __forceinline__ float skip_to(const MarchCfg& c, float t, float tt) {   // :396-398
    if (c.dt_gamma == 0.0f && t > 0.0f) {                  // (the first condition is launch-uniform)
        const float dt = clampf(0.0f, c.dt_min, c.dt_max);   // clamp(t * 0, dt_min, dt_max): dt_max when max_steps is so small that dt_min > dt_max
        int e;
        (void)frexpf(t, &e);                               // t in [2^(e-1), 2^e)
        const float top = scalbnf(1.0f, e), half_ulp = scalbnf(1.0f, e - 25);
        const float q = (t + dt) - t;                      // fl(t + dt) - t: exact, and the same for every T_k of the binade
        const float span = tt - t;
        const float k = fmaxf(ceilf(span * __builtin_amdgcn_rcpf(q)), 1.0f);   // within one of the true count (fixed below)
        if (q > 0.0f && fabsf(dt - q) != half_ulp && k <= 256.0f && (top - t) > (k + 2.0f) * q) {
            float cand = fmaf(k, q, t);                    // exact: k q and the sum are representable below 2^e
            if (cand < tt) cand += q;                      // k one short
            else if (k > 1.0f && cand - q >= tt) cand -= q;   // k one long
            return cand;
        }
    }
    do { t += clampf(t * c.dt_gamma, c.dt_min, c.dt_max); } while (t < tt);
    return t;
}

__device__ __forceinline__ Ray load_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, uint32_t i) {
    Ray r;
    r.ox = rays_o[3 * (size_t)i]; r.oy = rays_o[3 * (size_t)i + 1]; r.oz = rays_o[3 * (size_t)i + 2];
    r.dx = rays_d[3 * (size_t)i]; r.dy = rays_d[3 * (size_t)i + 1]; r.dz = rays_d[3 * (size_t)i + 2];
    r.rdx = 1.0f / r.dx; r.rdy = 1.0f / r.dy; r.rdz = 1.0f / r.dz;
    return r;
}

static MarchCfg make_cfg(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    MarchCfg c;
    const float SQRT3 = 1.7320508075688772f;
    c.bound = bound; c.dt_gamma = dt_gamma;
    c.dt_min = 2 * SQRT3 / (float)max_steps;
    c.dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / (float)H;
    c.rH = 1.0f / (float)H; c.Hf = (float)H; c.Cf = (float)C; c.Hm1 = (float)(H - 1);
    c.H3 = H * H * H;
    int e = 0;
    c.bound_exp = (bound > 0.0f && frexpf(bound, &e) == 0.5f) ? e - 1 : INT_MIN;
    return c;
}

__device__ __forceinline__ float step_of(const MarchCfg& c, float t) { return clampf(t * c.dt_gamma, c.dt_min, c.dt_max); }

// Constant step (dt_gamma == 0, every shipped config): inside one binade [2^(e-1), 2^e) all t are multiples of
// u = 2^(e-25+1) and fl(t + dt) = t + q with q = dt rounded to a multiple of u -- the SAME q for every t of the binade
// unless dt lies exactly half-way between two multiples (ties-to-even would then depend on t).  Hence
// T_k = T_0 + k*q exactly (k*q and the sum are representable while the sum stays below 2^e), and the 63-step serial
// recurrence collapses to one multiply-add per lane with bit-identical results.
__device__ __forceinline__ bool uniform_step(float t_base, float dt, float& q) {
    if (!(t_base > 0.0f)) return false;
    int e;
    (void)frexpf(t_base, &e);                               // t_base in [2^(e-1), 2^e)
    const float top = scalbnf(1.0f, e), half_ulp = scalbnf(1.0f, e - 25);
    q = (t_base + dt) - t_base;                             // exact when t_base + dt stays in the binade (checked below)
    const float r = dt - q;                                 // exact (Sterbenz)
    return (top - t_base) > 64.0f * q && fabsf(r) != half_ulp && q > 0.0f;
}

// 64 candidates starting at t_base: lane i gets T_{base+i}, identical rounding to the serial walk
__device__ __forceinline__ float candidate_t(const MarchCfg& cfg, float t_base, int lane) {
    float t = t_base, q;
    if (cfg.dt_gamma == 0.0f && uniform_step(t_base, clampf(0.0f, cfg.dt_min, cfg.dt_max), q)) {   // (not dt_min: the clamp yields dt_max when dt_min > dt_max)
        t = t_base + (float)lane * q;                     // exact, see uniform_step
    } else {
#pragma unroll 8
        for (int j = 0; j < 63; j++) {
            const float tn = t + step_of(cfg, t);
            t = (lane > j) ? tn : t;
        }
    }
    return t;
}

// ---------------------------------------------------------------- K1
// raymarching.cu:91-145
__global__ void k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                           const float* __restrict__ aabb, uint32_t N, float min_near,
                           float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float BIG = 3.402823466e+38f;
    float tn = 0.f, tf = 0.f;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (miss) break;
        const float o = rays_o[3 * (size_t)n + a];
        const float r = 1.0f / rays_d[3 * (size_t)n + a];
        float lo = (aabb[a] - o) * r, hi = (aabb[a + 3] - o) * r;
        if (lo > hi) { float s = lo; lo = hi; hi = s; }
        if (a == 0) { tn = lo; tf = hi; }
        else {
            if (tn > hi || lo > tf) { miss = true; }
            else { if (lo > tn) tn = lo; if (hi < tf) tf = hi; }
        }
    }
    if (miss) { nears[n] = BIG; fars[n] = BIG; return; }
    if (tn < min_near) tn = min_near;
    nears[n] = tn; fars[n] = tf;
}

#define STREAM(s) (reinterpret_cast<hipStream_t>(s))

}  // namespace
