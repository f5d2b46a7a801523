// raymarching.hip -- occupancy-grid ray marching + alpha compositing for gfx950.
//
// Replaces raymarching/src/raymarching.cu of the reference (cited per kernel).
// Written for wave64: ray compaction uses wavefront scans instead of the
// reference's two global atomicAdd reservations, which also makes the sample
// layout deterministic (ray-id order).
//
// Arithmetic contract (shared with oracle/lae_oracle.c): compiled with
// -ffp-contract=off; the a*b+c shapes that nvcc contracts in the reference are
// explicit fmaf() here, everything else is separate IEEE ops, so t-sequences,
// voxel indices, sample counts and positions are bit-identical to the oracle.
#include <string.h>
#include <algorithm>
#include <climits>
#include "lae_common.h"
#include "raymarch_common.h"

namespace {

// ---------------------------------------------------------------- K2
// raymarching.cu:162-198
__global__ void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                               float radius, uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float RPI = 0.3183098861837907f;
    const float ox = rays_o[3 * (size_t)n], oy = rays_o[3 * (size_t)n + 1], oz = rays_o[3 * (size_t)n + 2];
    const float dx = rays_d[3 * (size_t)n], dy = rays_d[3 * (size_t)n + 1], dz = rays_d[3 * (size_t)n + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float Bh = ox * dx + oy * dy + oz * dz;
    const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    coords[2 * (size_t)n] = 2 * atan2f(sqrtf(x * x + z * z), y) * RPI - 1;
    coords[2 * (size_t)n + 1] = atan2f(z, x) * RPI;
}

// ---------------------------------------------------------------- get_rays (nerf/utils.py:61-153)
// Pinhole rays of B cam2world poses for N pixel indices each (inds == NULL: every pixel in row-major order), optionally
// with the ray/box interval of K1 in the same pass (aabb != NULL).  The reference builds two H*W meshgrids, gathers them,
// stacks, normalises and runs a [N,3] x [3,3] matmul per call (~15 launches); pixel centre = index + 0.5 (:82-83).
__global__ void k_get_rays(const float* __restrict__ poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t W,
                           const int64_t* __restrict__ inds, uint64_t inds_stride, uint32_t N, float off_x, float off_y,
                           int perturb, float* __restrict__ rays_o, float* __restrict__ rays_d,
                           const float* __restrict__ aabb, float min_near, float* __restrict__ nears,
                           float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const float* __restrict__ P = poses + 16 * (size_t)b;
    const int64_t pix = inds ? inds[(size_t)b * inds_stride + n] : (int64_t)n;
    float i = (float)(uint32_t)(pix % W) + 0.5f, j = (float)(uint32_t)(pix / W) + 0.5f;
    if (perturb) { i -= off_x; j -= off_y; }                 // :133-136
    const float xs = (i - cx) / fx, ys = (j - cy) / fy;      // :138-139 (zs = 1)
    const float nrm = sqrtf(fmaf(ys, ys, xs * xs) + 1.0f);   // :141
    const float dx = xs / nrm, dy = ys / nrm, dz = 1.0f / nrm;
    float d[3], o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {                            // :142 directions @ R^T, :144 translation column
        d[k] = fmaf(dz, P[4 * k + 2], fmaf(dy, P[4 * k + 1], dx * P[4 * k]));
        o[k] = P[4 * k + 3];
    }
    const size_t r = (size_t)b * N + n;
#pragma unroll
    for (int k = 0; k < 3; k++) { rays_o[3 * r + k] = o[k]; rays_d[3 * r + k] = d[k]; }
    if (!aabb) return;
    const float BIG = 3.402823466e+38f;                      // K1, raymarching.cu:91-145
    float tn = 0.f, tf = 0.f;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (miss) break;
        const float rcp = 1.0f / d[a];
        float lo = (aabb[a] - o[a]) * rcp, hi = (aabb[a + 3] - o[a]) * rcp;
        if (lo > hi) { float sw = lo; lo = hi; hi = sw; }
        if (a == 0) { tn = lo; tf = hi; }
        else {
            if (tn > hi || lo > tf) { miss = true; }
            else { if (lo > tn) tn = lo; if (hi < tf) tf = hi; }
        }
    }
    if (miss) { nears[r] = BIG; fars[r] = BIG; return; }
    if (tn < min_near) tn = min_near;
    nears[r] = tn; fars[r] = tf;
}

// ---------------------------------------------------------------- K3 / K4 / K5
__global__ void k_morton3D(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton_encode((uint32_t)coords[3 * (size_t)n], (uint32_t)coords[3 * (size_t)n + 1],
                                        (uint32_t)coords[3 * (size_t)n + 2]);
}
__global__ void k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t ind = indices[n];
    coords[3 * (size_t)n + 0] = (int32_t)morton_compact((uint32_t)(ind >> 0));
    coords[3 * (size_t)n + 1] = (int32_t)morton_compact((uint32_t)(ind >> 1));
    coords[3 * (size_t)n + 2] = (int32_t)morton_compact((uint32_t)(ind >> 2));
}
// raymarching.cu:267-289; one thread per output byte, two 16-byte loads
__global__ void k_packbits(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[2 * (size_t)n];
    const float4 b = reinterpret_cast<const float4*>(grid)[2 * (size_t)n + 1];
    uint32_t bits = (a.x > thresh) | ((a.y > thresh) << 1) | ((a.z > thresh) << 2) | ((a.w > thresh) << 3) |
                    ((b.x > thresh) << 4) | ((b.y > thresh) << 5) | ((b.z > thresh) << 6) | ((b.w > thresh) << 7);
    bitfield[n] = (uint8_t)bits;
}

// ---------------------------------------------------------------- K6 (training march)
// raymarching.cu:311-480.  MI355X form: ONE WAVEFRONT PER RAY.
//
// The reference walks a ray with one thread: probe the bitfield (a dependent 1-byte load), then either emit a
// sample and advance by dt, or skip to the voxel exit with `do t += dt while (t < tt)`.  Either way t only ever
// moves along the fixed sequence T_{k+1} = T_k + clamp(T_k * dt_gamma, dt_min, dt_max): occupancy decides
// which T_k are VISITED, never their values.  So a wave evaluates 64 consecutive candidates at once
//   1. lanes build T_k..T_k+63 with the exact serial fp32 recurrence (pure VALU, no memory),
//   2. every lane probes its candidate (64 bitfield loads in flight instead of 1),
//   3. the visited chain is resolved on the scalar unit from two ballots: runs of occupied candidates are
//      taken whole, an empty candidate jumps to the first T_j >= tt (ballot + ctz),
//   4. emitted samples are ranked with popcounts of the emit mask -> consecutive lanes write consecutive rows.
// Counts, offsets and every emitted float are bit-identical to the serial walk (same probe_at / same sums).
// cooperative zero fill of sample rows [rows_end, M): wave n of N takes the n-th slice
__device__ __forceinline__ void zero_tail_rows(uint32_t rows_end, uint32_t M, uint32_t n, uint32_t N, int lane, float* __restrict__ buf,
                                               uint32_t floats_per_row) {
    if (rows_end >= M) return;
    const uint32_t tail = M - rows_end, chunk = (tail + N - 1) / N;
    const uint32_t lo = rows_end + min(tail, n * chunk), hi = rows_end + min(tail, (n + 1) * chunk);
    for (size_t e = (size_t)lo * floats_per_row + lane; e < (size_t)hi * floats_per_row; e += 64) buf[e] = 0.0f;
}

constexpr int MARCH_WAVES = 4;                 // rays per 256-thread block
constexpr int MARCH_BLOCK = 64 * MARCH_WAVES;


// The count pass leaves one record per candidate chunk that emitted samples: where the chunk starts and which lanes
// emit.  The emit pass replays the records (no bitfield probes, no chain resolution): it only rebuilds the candidate
// times and writes rows.  A ray with more than MARCH_REC_MAX such chunks is flagged and walked again instead.
constexpr uint32_t MARCH_REC_MAX = 24;
constexpr uint32_t MARCH_REC_OVERFLOW = 0xffffffffu;
struct MarchRec { float t_base; uint32_t lo, hi; };
struct MarchRecs { uint32_t* nrec; MarchRec* rec; };      // nrec[N], rec[N * MARCH_REC_MAX]; both NULL = feature off

// EMIT == false: count the samples of ray n.  EMIT == true: write them at `offset` (num_steps known).
template <bool EMIT>
__global__ __launch_bounds__(MARCH_BLOCK) void k_march_train_wave(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, MarchCfg cfg,
    uint32_t max_steps, uint32_t N, uint32_t M, const float* __restrict__ nears, const float* __restrict__ fars,
    const float* __restrict__ noises, uint32_t* __restrict__ counts, const uint32_t* __restrict__ prefix,
    float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas, int32_t* __restrict__ rays, MarchRecs recs) {
    const uint32_t n = blockIdx.x * MARCH_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // one ray per wave: scalar
    if (n >= N) return;                                   // whole wave
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t n_rec = 0;

    uint32_t limit = max_steps, offset = 0;
    if (EMIT) {
        const uint32_t rows_end = prefix[N + 2];
        zero_tail_rows(rows_end, M, n, N, lane, xyzs, 3);
        zero_tail_rows(rows_end, M, n, N, lane, dirs, 3);
        zero_tail_rows(rows_end, M, n, N, lane, deltas, 2);
        limit = counts[n];
        offset = prefix[N] + prefix[n];                   // prefix[N] = counter value before this call (:405)
        if (lane == 0) {
            const uint32_t row = prefix[N + 1] + n;       // :406 (ray-id order)
            rays[3 * (size_t)row + 0] = (int32_t)n;
            rays[3 * (size_t)row + 1] = (int32_t)offset;
            rays[3 * (size_t)row + 2] = (int32_t)limit;
        }
        if (limit == 0 || offset + limit > M) return;     // :415-416 (overflowing rays write nothing)
    }
    const Ray r = load_ray(rays_o, rays_d, n);
    const float far = fars[n];
    float t_base = nears[n];
    t_base = fmaf(step_of(cfg, t_base), noises[n], t_base);                   // :351
    float last_t = t_base;                                // post-step t of the previous emitted sample
    uint32_t emitted = 0;
    bool pending = false;                                 // walker is skipping towards pending_tt
    float pending_tt = 0.f;

    if (EMIT && recs.nrec && recs.nrec[n] != MARCH_REC_OVERFLOW) {
        // replay: same candidate times, same emit masks, same row arithmetic as the walk below
        const uint32_t nr = recs.nrec[n];
        const MarchRec* __restrict__ rr = recs.rec + (size_t)n * MARCH_REC_MAX;
        for (uint32_t k = 0; k < nr; k++) {
            const MarchRec rc = rr[k];
            const unsigned long long emit = (unsigned long long)rc.lo | ((unsigned long long)rc.hi << 32);
            const float t = candidate_t(cfg, rc.t_base, lane);
            const float dt = step_of(cfg, t);
            const float t_next = t + dt;
            const unsigned long long pm = emit & below;
            const int prev_lane = pm ? 63 - __builtin_clzll(pm) : 0;
            const float prev_next = __shfl(t_next, prev_lane, 64);
            if ((emit >> lane) & 1ull) {
                const size_t row = (size_t)offset + emitted + (uint32_t)__builtin_popcountll(pm);
                float* px = xyzs + 3 * row; float* pd = dirs + 3 * row; float* pl = deltas + 2 * row;
                px[0] = clampf(fmaf(t, r.dx, r.ox), -cfg.bound, cfg.bound);
                px[1] = clampf(fmaf(t, r.dy, r.oy), -cfg.bound, cfg.bound);
                px[2] = clampf(fmaf(t, r.dz, r.oz), -cfg.bound, cfg.bound);
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                pl[0] = dt;
                pl[1] = t_next - (pm ? prev_next : last_t);                   // :461
            }
            const int top = 63 - __builtin_clzll(emit);
            last_t = __shfl(t_next, top, 64);
            emitted += (uint32_t)__builtin_popcountll(emit);
        }
        return;
    }

    while (t_base < far && emitted < limit) {
        // 1. candidates: lane i gets T_{base+i} by i serial steps (identical rounding to the serial walk)
        const float t = candidate_t(cfg, t_base, lane);
        const float dt = step_of(cfg, t);
        const float t_next = t + dt;                      // == T_{base+i+1}
        const bool valid = t < far;
        // 2. probe
        Probe p;
        p.occ = false; p.tt = t; p.x = p.y = p.z = 0.f; p.dt = dt; p.index = 0;
        if (valid) p = probe_at(r, cfg, grid, t);
        const unsigned long long valid_mask = __ballot(valid);
        const unsigned long long occ_mask = __ballot(valid && p.occ);
        // 3. visited chain (all scalar)
        unsigned long long emit = 0ull;
        int k = 0;
        if (pending) {
            const unsigned long long m = __ballot(valid && t >= pending_tt);
            if (m) { k = __builtin_ctzll(m); pending = false; } else k = 64;
        }
        while (k < 64 && ((valid_mask >> k) & 1ull)) {
            if ((occ_mask >> k) & 1ull) {
                const unsigned long long inv = (~occ_mask) >> k;
                const int run = inv ? __builtin_ctzll(inv) : 64 - k;          // bits beyond `valid` are 0 in occ_mask
                emit |= ((run >= 64) ? ~0ull : ((1ull << run) - 1ull)) << k;
                k += run;
            } else {
                const float tt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p.tt), k));
                unsigned long long m = __ballot(valid && t >= tt);
                m &= (k >= 63) ? 0ull : ~((2ull << k) - 1ull);                // at least one step (:396-398)
                if (m) k = __builtin_ctzll(m);
                else { pending = true; pending_tt = tt; k = 64; }
            }
        }
        // cap at `limit` samples (:359 / :427)
        uint32_t cnt = (uint32_t)__builtin_popcountll(emit);
        bool done = false;
        if (emitted + cnt > limit) {
            const uint32_t keep = limit - emitted;
            const bool mine = ((emit >> lane) & 1ull) && (uint32_t)__builtin_popcountll(emit & below) < keep;
            emit = __ballot(mine);
            cnt = keep;
            done = true;
        }
        if (!EMIT && recs.nrec && emit) {                 // leave a record for the emit pass
            if (n_rec < MARCH_REC_MAX) {
                if (lane == 0) recs.rec[(size_t)n * MARCH_REC_MAX + n_rec] = MarchRec{t_base, (uint32_t)emit, (uint32_t)(emit >> 32)};
                n_rec++;
            } else n_rec = MARCH_REC_OVERFLOW;
        }
        // 4. emit
        if (EMIT && emit) {
            const unsigned long long pm = emit & below;
            const int prev_lane = pm ? 63 - __builtin_clzll(pm) : 0;
            const float prev_next = __shfl(t_next, prev_lane, 64);
            if ((emit >> lane) & 1ull) {
                const size_t row = (size_t)offset + emitted + (uint32_t)__builtin_popcountll(pm);
                float* px = xyzs + 3 * row; float* pd = dirs + 3 * row; float* pl = deltas + 2 * row;
                px[0] = p.x; px[1] = p.y; px[2] = p.z;
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                pl[0] = dt;
                pl[1] = t_next - (pm ? prev_next : last_t);                   // :461
            }
            const int top = 63 - __builtin_clzll(emit);
            last_t = __shfl(t_next, top, 64);
        }
        emitted += cnt;
        if (done || valid_mask != ~0ull) break;           // cap reached, or the ray left [near, far) in this chunk
        t_base = __shfl(t_next, 63, 64);
    }
    if (!EMIT && lane == 0) {
        counts[n] = emitted;
        if (recs.nrec) recs.nrec[n] = n_rec;
    }
}

// exclusive scan of the per-ray counts (single block) + counter update (:405-406).
// prefix[0..N) = exclusive prefix, prefix[N] / prefix[N+1] = counter values before this call.
// prefix[N + 2] = rows_end: first sample row not written by this call (end of the last ray that fits into M);
// the emit pass zero-fills [rows_end, M) so callers need not pre-zero xyzs / dirs / deltas.
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* __restrict__ counts, uint32_t N, uint32_t M,
                                                       uint32_t* __restrict__ prefix, int32_t* __restrict__ counter,
                                                       uint32_t* __restrict__ rows_end_out) {
    __shared__ uint32_t lds[17];
    __shared__ uint32_t s_end, s_base;
    if (threadIdx.x == 0) { s_end = 0; s_base = counter ? (uint32_t)counter[0] : 0u; }
    __syncthreads();
    const uint32_t b0u = s_base;
    uint32_t carry = 0, my_end = 0;
    for (uint32_t base = 0; base < N; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < N ? counts[i] : 0;
        uint32_t total;
        const uint32_t ex = lae::block_excl_scan<16>(v, &total, lds);
        if (i < N) {
            prefix[i] = carry + ex;
            const unsigned long long end = (unsigned long long)b0u + carry + ex + v;
            if (v != 0 && end <= (unsigned long long)M) my_end = max(my_end, carry + ex + v);
        }
        carry += total;
    }
    atomicMax(&s_end, my_end);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t b0 = 0, b1 = 0;
        if (counter) {
            b0 = atomicAdd(counter, (int32_t)carry);
            b1 = atomicAdd(counter + 1, (int32_t)N);
        }
        prefix[N] = (uint32_t)b0;
        prefix[N + 1] = (uint32_t)b1;
        prefix[N + 2] = min(M, (uint32_t)b0 + s_end);
        if (rows_end_out) rows_end_out[0] = prefix[N + 2];
    }
}


// ---------------------------------------------------------------- K7 / K8 (training composite)
// raymarching.cu:500-577 / :601-682.  MI355X form: one wavefront per ray, 64 samples per pass.
// Transmittance T_k = prod_{j<k} (1 - alpha_j) is a wave-level multiplicative scan, the running colour /
// depth sums are additive scans, the early stop `T < T_thresh` is a ballot + ctz.  Sample reads and gradient
// writes are lane-consecutive (coalesced); the reference's one-thread-per-ray loop strides by ray.
// Scan association differs from the serial loop -> results agree to fp32 rounding (tests: 2e-6).
// Wave-level scans on the DPP data path (row shifts 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 / :31 across
// them -- GFX9 controls, present on gfx950): six dependent VALU operations per scan.  Written with __shfl_up the compiler
// emits ds_bpermute_b32, an LDS round trip per step: 74 of them in the training compositing kernel, whose duration is the
// dependent chain of its longest rays (several passes of 64 samples, forward and backward).  Lanes that a shift leaves
// without a source take the operation's identity (`old` operand), so no per-step select is needed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_scan_add(float v, int) {
    v += dpp_f<0x111, 0xf>(0.0f, v); v += dpp_f<0x112, 0xf>(0.0f, v); v += dpp_f<0x114, 0xf>(0.0f, v); v += dpp_f<0x118, 0xf>(0.0f, v);
    v += dpp_f<0x142, 0xa>(0.0f, v);        // row_bcast:15 -> rows 1 and 3
    v += dpp_f<0x143, 0xc>(0.0f, v);        // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v, int) {
    v *= dpp_f<0x111, 0xf>(1.0f, v); v *= dpp_f<0x112, 0xf>(1.0f, v); v *= dpp_f<0x114, 0xf>(1.0f, v); v *= dpp_f<0x118, 0xf>(1.0f, v);
    v *= dpp_f<0x142, 0xa>(1.0f, v);
    v *= dpp_f<0x143, 0xc>(1.0f, v);
    return v;
}
__device__ __forceinline__ float wave_last(float v) {        // value of lane 63 in every lane
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_prev(float v, float first) {   // value of the lane below; `first` in lane 0 (wave_shr:1)
    return dpp_f<0x138, 0xf>(first, v);
}
__device__ __forceinline__ float wave_sum(float v) { return wave_last(wave_scan_add(v, 0)); }

constexpr int COMP_WAVES = 4;
constexpr int COMP_BLOCK = 64 * COMP_WAVES;

// BLEND: additionally the post-ops of run_cuda (renderer.py:321, 325) in the epilogue:
//   image_out = image + (1 - weights_sum) * bg,  depth_out = clamp(depth - near, min=0) / (far - near);
// `image` keeps the un-blended colour the backward needs.
struct Blend { const float* nears; const float* fars; const float* bg_rays; float bg[3]; float* image_out; float* depth_out; };

template <bool BLEND>
__global__ __launch_bounds__(COMP_BLOCK) void k_composite_train_fwd(
    const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
    const int32_t* __restrict__ rays, uint32_t M, uint32_t N, float T_thresh, float* __restrict__ weights_sum,
    float* __restrict__ depth, float* __restrict__ image, Blend bl) {
    const uint32_t n = blockIdx.x * COMP_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // one ray per wave: scalar
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1];
    const uint32_t num_steps = (uint32_t)rays[3 * (size_t)n + 2];
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (!(num_steps == 0 || offset + num_steps > M)) {                      // :521
        float T = 1.0f, t = 0.0f;
        for (uint32_t base = 0; base < num_steps; base += 64) {
            const uint32_t k = base + lane;
            bool valid = k < num_steps;
            float alpha = 0.f, d1 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (valid) {
                const size_t i = (size_t)offset + k;
                alpha = 1.0f - __expf(-sigmas[i] * deltas[2 * i]);
                d1 = deltas[2 * i + 1];
                c0 = rgbs[3 * i]; c1 = rgbs[3 * i + 1]; c2 = rgbs[3 * i + 2];
            }
            const float incl = wave_scan_mul(1.0f - alpha, lane);           // prod_{j<=k} within the pass
            const float excl = wave_prev(incl, 1.0f);
            const float T_post = T * incl;
            const unsigned long long stop = __ballot(valid && T_post < T_thresh);   // :557 (sample included)
            bool done = false;
            if (stop) { const int last = __builtin_ctzll(stop); valid = valid && lane <= last; done = true; }
            const float w = valid ? alpha * (T * excl) : 0.0f;
            const float tk = t + wave_scan_add(d1, lane);
            r += wave_sum(w * c0); g += wave_sum(w * c1); b += wave_sum(w * c2);
            d += wave_sum(w * tk); ws += wave_sum(w);
            if (done) break;
            T *= wave_last(incl);
            t = wave_last(tk);
        }
    }
    if (lane == 0) {
        weights_sum[index] = ws; depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
        if constexpr (BLEND) {
            const float* bg = bl.bg_rays ? bl.bg_rays + 3 * (size_t)index : bl.bg;
            const float rest = 1.0f - ws;
            bl.image_out[3 * (size_t)index] = r + rest * bg[0];
            bl.image_out[3 * (size_t)index + 1] = g + rest * bg[1];
            bl.image_out[3 * (size_t)index + 2] = b + rest * bg[2];
            const float nr = bl.nears[index];
            bl.depth_out[index] = fmaxf(d - nr, 0.0f) / (bl.fars[index] - nr);
        }
    }
}

// DENSE: (a) the gradient of the BLEND epilogue is folded in: grad_ws_eff = grad_ws - sum_c grad_image_c * bg_c;
// (b) EVERY row of grad_sigmas / grad_rgbs in [0, M) is written (zeros for samples after the early stop and for the
// rows [rows_end, M) no ray owns), so the caller allocates them uninitialised.  Needs the contiguous ray-id-order
// layout lae_march_rays_train produces.
struct Dense { const float* bg_rays; float bg[3]; const uint32_t* rows_end; const float* grad_scale; };   // grad_scale: device scalar or NULL

template <bool DENSE>
__global__ __launch_bounds__(COMP_BLOCK) void k_composite_train_bwd(
    const float* __restrict__ grad_ws, const float* __restrict__ grad_image, const float* __restrict__ sigmas,
    const float* __restrict__ rgbs, const float* __restrict__ deltas, const int32_t* __restrict__ rays,
    const float* __restrict__ weights_sum, const float* __restrict__ image, uint32_t M, uint32_t N, float T_thresh,
    float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs, Dense dn) {
    const uint32_t n = blockIdx.x * COMP_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // one ray per wave: scalar
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    if constexpr (DENSE) {
        const uint32_t rows_end = dn.rows_end[0];
        zero_tail_rows(rows_end, M, n, N, lane, grad_sigmas, 1);
        zero_tail_rows(rows_end, M, n, N, lane, grad_rgbs, 3);
    }
    const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1];
    const uint32_t num_steps = (uint32_t)rays[3 * (size_t)n + 2];
    if (num_steps == 0 || offset + num_steps > M) return;                   // :624
    float gws = grad_ws ? grad_ws[index] : 0.0f;
    float g0 = grad_image[3 * (size_t)index], g1 = grad_image[3 * (size_t)index + 1], g2 = grad_image[3 * (size_t)index + 2];
    if constexpr (DENSE) {
        if (dn.grad_scale) {                               // fused criterion: upstream d(loss) arrives as a device scalar
            const float gs = dn.grad_scale[0];
            g0 *= gs; g1 *= gs; g2 *= gs; gws *= gs;
        }
        const float* bg = dn.bg_rays ? dn.bg_rays + 3 * (size_t)index : dn.bg;
        gws = gws - ((g0 * bg[0] + g1 * bg[1]) + g2 * bg[2]);
    }
    const float rf = image[3 * (size_t)index], gf = image[3 * (size_t)index + 1], bf = image[3 * (size_t)index + 2];
    const float tail = gws * (1 - weights_sum[index]);
    float T = 1.0f, r = 0, g = 0, b = 0;
    bool stopped = false;
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t k = base + lane;
        bool valid = k < num_steps;
        const size_t i = (size_t)offset + k;
        if (DENSE && stopped) {                                             // samples after the early stop: zero gradient
            if (valid) { grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f; grad_sigmas[i] = 0.f; }
            continue;
        }
        const bool in_ray = valid;
        float alpha = 0.f, d0 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (valid) {
            d0 = deltas[2 * i];
            alpha = 1.0f - __expf(-sigmas[i] * d0);
            c0 = rgbs[3 * i]; c1 = rgbs[3 * i + 1]; c2 = rgbs[3 * i + 2];
        }
        const float incl = wave_scan_mul(1.0f - alpha, lane);
        const float excl = wave_prev(incl, 1.0f);
        const float T_post = T * incl;
        const unsigned long long stop = __ballot(valid && T_post < T_thresh);
        bool done = false;
        if (stop) { const int last = __builtin_ctzll(stop); valid = valid && lane <= last; done = true; }
        const float w = valid ? alpha * (T * excl) : 0.0f;
        const float rk = r + wave_scan_add(w * c0, lane);                   // running sums INCLUDING this sample
        const float gk = g + wave_scan_add(w * c1, lane);
        const float bk = b + wave_scan_add(w * c2, lane);
        if (valid) {
            grad_rgbs[3 * i] = g0 * w; grad_rgbs[3 * i + 1] = g1 * w; grad_rgbs[3 * i + 2] = g2 * w;      // :657-659
            grad_sigmas[i] = d0 * (g0 * (T_post * c0 - (rf - rk)) + g1 * (T_post * c1 - (gf - gk)) +
                                   g2 * (T_post * c2 - (bf - bk)) + tail);                                // :662-667
        } else if (DENSE && in_ray) {
            grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f; grad_sigmas[i] = 0.f;
        }
        if (done) { if (DENSE) { stopped = true; continue; } break; }
        T *= wave_last(incl);
        r = wave_last(rk); g = wave_last(gk); b = wave_last(bk);
    }
}

// ---------------------------------------------------------------- K7 + criterion + K8 in one launch (training step)
// composite_rays_train forward with the BLEND epilogue, the trainer's MSE criterion (d loss / d pixel needs this ray's
// pixel only) and composite_rays_train backward (DENSE), one wavefront per ray: the three launches of the step's middle
// (11.8 + 5.6 + 11 us and two kernel boundaries) become one.  Arithmetic is that of k_composite_train_fwd<true>,
// k_mse_fwd and k_composite_train_bwd<true> (grad_scale = 1), statement for statement.  The loss VALUE needs all rays:
// every workgroup leaves the sum of its rays' squared errors in partials[blockIdx.x], k_loss_finish adds them in a fixed
// order (deterministic).
struct StepLoss { const float* target; const float* scale; float* grad_image; float* partials; float* poison_loss; };

__global__ __launch_bounds__(COMP_BLOCK) void k_composite_train_step(
    const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
    const int32_t* __restrict__ rays, uint32_t M, uint32_t N, float T_thresh, float* __restrict__ weights_sum,
    float* __restrict__ depth, float* __restrict__ image, Blend bl, StepLoss sl, const uint32_t* __restrict__ rows_end_p,
    float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    __shared__ float s_sq[COMP_WAVES];
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n = blockIdx.x * COMP_WAVES + wv;                      // one ray per wave: scalar
    const int lane = threadIdx.x & 63;
    float sq = 0.0f;
    if (n < N) {
        const uint32_t rows_end = rows_end_p[0];
        zero_tail_rows(rows_end, M, n, N, lane, grad_sigmas, 1);
        zero_tail_rows(rows_end, M, n, N, lane, grad_rgbs, 3);
        const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1];
        const uint32_t num_steps = (uint32_t)rays[3 * (size_t)n + 2];
        const bool has = !(num_steps == 0 || offset + num_steps > M);
        // ---- forward (k_composite_train_fwd<true>)
        float r = 0, g = 0, b = 0, ws = 0, d = 0;
        // operands of one pass of 64 samples; the next pass is requested before the current one is scanned: the kernel's
        // time is that of its longest rays (several passes, forward and backward), one memory latency per pass otherwise
        struct Pass { float sg, d0, d1, c0, c1, c2; };
        auto fetch = [&](uint32_t base) {
            Pass p{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const uint32_t k = base + lane;
            if (k < num_steps) {
                const size_t i = (size_t)offset + k;
                p.sg = sigmas[i]; p.d0 = deltas[2 * i]; p.d1 = deltas[2 * i + 1];
                p.c0 = rgbs[3 * i]; p.c1 = rgbs[3 * i + 1]; p.c2 = rgbs[3 * i + 2];
            }
            return p;
        };
        if (has) {
            float T = 1.0f, t = 0.0f;
            Pass nxt = fetch(0);
            for (uint32_t base = 0; base < num_steps; base += 64) {
                const uint32_t k = base + lane;
                bool valid = k < num_steps;
                const Pass cur = nxt;
                if (base + 64 < num_steps) nxt = fetch(base + 64);
                float alpha = 0.f, d1 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
                if (valid) {
                    alpha = 1.0f - __expf(-cur.sg * cur.d0);
                    d1 = cur.d1;
                    c0 = cur.c0; c1 = cur.c1; c2 = cur.c2;
                }
                const float incl = wave_scan_mul(1.0f - alpha, lane);
                const float excl = wave_prev(incl, 1.0f);
                const float T_post = T * incl;
                const unsigned long long stop = __ballot(valid && T_post < T_thresh);
                bool done = false;
                if (stop) { const int last = __builtin_ctzll(stop); valid = valid && lane <= last; done = true; }
                const float w = valid ? alpha * (T * excl) : 0.0f;
                const float tk = t + wave_scan_add(d1, lane);
                r += wave_sum(w * c0); g += wave_sum(w * c1); b += wave_sum(w * c2);
                d += wave_sum(w * tk); ws += wave_sum(w);
                if (done) break;
                T *= wave_last(incl);
                t = wave_last(tk);
            }
        }
        const float* bg = bl.bg_rays ? bl.bg_rays + 3 * (size_t)index : bl.bg;
        const float rest = 1.0f - ws;
        const float o0 = r + rest * bg[0], o1 = g + rest * bg[1], o2 = b + rest * bg[2];
        // ---- criterion (k_mse_fwd): grad = ((pred - target) * 2 / n_elements) * scale
        const float s = sl.scale ? sl.scale[0] : 1.0f;
        const float gk = 2.0f / (float)(3u * N);
        const float e0 = o0 - sl.target[3 * (size_t)index], e1 = o1 - sl.target[3 * (size_t)index + 1],
                    e2 = o2 - sl.target[3 * (size_t)index + 2];
        sq = fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
        float g0 = (e0 * gk) * s, g1 = (e1 * gk) * s, g2 = (e2 * gk) * s;
        if (lane == 0) {
            weights_sum[index] = ws; depth[index] = d;
            image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
            bl.image_out[3 * (size_t)index] = o0; bl.image_out[3 * (size_t)index + 1] = o1; bl.image_out[3 * (size_t)index + 2] = o2;
            const float nr = bl.nears[index];
            bl.depth_out[index] = fmaxf(d - nr, 0.0f) / (bl.fars[index] - nr);
            sl.grad_image[3 * (size_t)index] = g0; sl.grad_image[3 * (size_t)index + 1] = g1; sl.grad_image[3 * (size_t)index + 2] = g2;
        }
        // ---- backward (k_composite_train_bwd<true>, grad_weights_sum = 0, grad_scale = 1)
        if (has) {
            const float gws = 0.0f - ((g0 * bg[0] + g1 * bg[1]) + g2 * bg[2]);
            const float rf = r, gf = g, bf = b;
            const float tail = gws * (1 - ws);
            float T = 1.0f, rr = 0, gg = 0, bb = 0;
            bool stopped = false;
            Pass nxt = fetch(0);
            for (uint32_t base = 0; base < num_steps; base += 64) {
                const uint32_t k = base + lane;
                bool valid = k < num_steps;
                const size_t i = (size_t)offset + k;
                const Pass cur = nxt;
                if (!stopped && base + 64 < num_steps) nxt = fetch(base + 64);
                if (stopped) {                                                      // samples after the early stop: zero gradient
                    if (valid) { grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f; grad_sigmas[i] = 0.f; }
                    continue;
                }
                const bool in_ray = valid;
                float alpha = 0.f, d0 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
                if (valid) {
                    d0 = cur.d0;
                    alpha = 1.0f - __expf(-cur.sg * d0);
                    c0 = cur.c0; c1 = cur.c1; c2 = cur.c2;
                }
                const float incl = wave_scan_mul(1.0f - alpha, lane);
                const float excl = wave_prev(incl, 1.0f);
                const float T_post = T * incl;
                const unsigned long long stop = __ballot(valid && T_post < T_thresh);
                bool done = false;
                if (stop) { const int last = __builtin_ctzll(stop); valid = valid && lane <= last; done = true; }
                const float w = valid ? alpha * (T * excl) : 0.0f;
                const float rk = rr + wave_scan_add(w * c0, lane);
                const float gkk = gg + wave_scan_add(w * c1, lane);
                const float bk = bb + wave_scan_add(w * c2, lane);
                if (valid) {
                    grad_rgbs[3 * i] = g0 * w; grad_rgbs[3 * i + 1] = g1 * w; grad_rgbs[3 * i + 2] = g2 * w;
                    grad_sigmas[i] = d0 * (g0 * (T_post * c0 - (rf - rk)) + g1 * (T_post * c1 - (gf - gkk)) +
                                           g2 * (T_post * c2 - (bf - bk)) + tail);
                } else if (in_ray) {
                    grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f; grad_sigmas[i] = 0.f;
                }
                if (done) { stopped = true; continue; }
                T *= wave_last(incl);
                rr = wave_last(rk); gg = wave_last(gkk); bb = wave_last(bk);
            }
        }
    }
    if (lane == 0) s_sq[wv] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < COMP_WAVES; w++) t += s_sq[w];
        sl.partials[blockIdx.x] = t;
        // deferred loss value (lae_composite_rays_train_step with poison_loss): whoever reads it before the finishing
        // launch (lae_loss_finish, or the extra block of lae_nerf_head_backward) sees NaN, not a stale number
        if (blockIdx.x == 0 && sl.poison_loss) { sl.poison_loss[0] = __builtin_nanf(""); sl.poison_loss[1] = __builtin_nanf(""); }
    }
}

// sum of the workgroups' squared-error sums in a fixed order -> loss_out[0] = mean * scale, loss_out[1] = mean
__global__ __launch_bounds__(1024) void k_loss_finish(const float* __restrict__ partials, uint32_t n_part, uint32_t n_elem,
                                                       const float* __restrict__ scale, float* __restrict__ loss_out) {
    __shared__ float part[16];
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < n_part; i += 1024) acc += partials[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; w++) t += part[w];
        const float loss = t / (float)n_elem;
        loss_out[0] = loss * (scale ? scale[0] : 1.0f);
        loss_out[1] = loss;
    }
}

// ---------------------------------------------------------------- K9 / K10 (inference march)
// raymarching.cu:700-805 and :811-926 (EDIT = distill variant)
template <bool EDIT>
__global__ void k_march_infer(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                              const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                              const float* __restrict__ rays_d, MarchCfg cfg, const uint8_t* __restrict__ grid,
                              const uint8_t* __restrict__ edit_grid, const float* __restrict__ fars,
                              float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                              uint8_t* __restrict__ edit_occ, const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t index = rays_alive[n];
    const Ray r = load_ray(rays_o, rays_d, (uint32_t)index);
    float* px = xyzs + 3 * (size_t)n * n_step;
    float* pd = dirs + 3 * (size_t)n * n_step;
    float* pl = deltas + 2 * (size_t)n * n_step;
    uint8_t* pe = EDIT ? edit_occ + (size_t)n * n_step : nullptr;
    float t = rays_t[index];
    const float far = fars[index];
    t = fmaf(clampf(t * cfg.dt_gamma, cfg.dt_min, cfg.dt_max), noises[n], t);     // :746
    float last_t = t;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        const Probe p = probe_at(r, cfg, grid, t);
        if (p.occ) {
            px[0] = p.x; px[1] = p.y; px[2] = p.z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            t += p.dt;
            pl[0] = p.dt; pl[1] = t - last_t; last_t = t;
            if (EDIT) {
                if ((edit_grid[p.index >> 3] >> (p.index & 7u)) & 1u) *pe = 1;
                pe++;
            }
            px += 3; pd += 3; pl += 2; step++;
        } else t = skip_to(cfg, t, p.tt);
    }
}

// ---------------------------------------------------------------- K11 / K12 (inference composite)
// raymarching.cu:948-1035 and :1037-1142
template <bool EDIT>
__global__ void k_composite_infer(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* __restrict__ rays_alive,
                                  float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                  const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                  float* __restrict__ weights_sum, float* __restrict__ weights_edit_sum,
                                  float* __restrict__ depth, float* __restrict__ depth_edit,
                                  const uint8_t* __restrict__ edit_occ, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t index = rays_alive[n];
    const float* s = sigmas + (size_t)n * n_step;
    const float* c = rgbs + 3 * (size_t)n * n_step;
    const float* dl = deltas + 2 * (size_t)n * n_step;
    const uint8_t* eo = EDIT ? edit_occ + (size_t)n * n_step : nullptr;
    float t = rays_t[index];
    float ws = weights_sum[index], d = depth[index];
    float wse = 0, de = 0;
    if (EDIT) { wse = weights_edit_sum[index]; de = depth_edit[index]; }
    float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
    uint32_t step = 0;
    while (step < n_step) {
        const float d0 = dl[2 * step];
        if (d0 == 0) break;
        const float alpha = 1.0f - __expf(-s[step] * d0);
        const float T = 1 - ws;
        const float w = alpha * T;
        ws += w;
        if (EDIT) { if (eo[step]) { wse += w; de = fmaf(w, t, de); } }
        t += dl[2 * step + 1];
        d = fmaf(w, t, d);
        r = fmaf(w, c[3 * step], r); g = fmaf(w, c[3 * step + 1], g); b = fmaf(w, c[3 * step + 2], b);
        if (T < T_thresh) break;
        step++;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = ws; depth[index] = d;
    if (EDIT) { weights_edit_sum[index] = wse; depth_edit[index] = de; }
    image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
}

// ---------------------------------------------------------------- alive-list compaction
constexpr int COMPACT_BLOCK = 256;
__global__ __launch_bounds__(COMPACT_BLOCK) void k_compact_count(const int32_t* __restrict__ rays_alive, uint32_t n,
                                                                  uint32_t* __restrict__ local_prefix,
                                                                  uint32_t* __restrict__ block_totals) {
    __shared__ uint32_t lds[COMPACT_BLOCK / 64 + 1];
    const uint32_t i = blockIdx.x * COMPACT_BLOCK + threadIdx.x;
    const uint32_t keep = (i < n && rays_alive[i] >= 0) ? 1u : 0u;
    uint32_t total;
    const uint32_t ex = lae::block_excl_scan<COMPACT_BLOCK / 64>(keep, &total, lds);
    if (i < n) local_prefix[i] = ex;
    if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t* __restrict__ totals, uint32_t nblk,
                                                        int32_t* __restrict__ n_out) {
    __shared__ uint32_t lds[17];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblk; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblk ? totals[i] : 0;
        uint32_t total;
        const uint32_t ex = lae::block_excl_scan<16>(v, &total, lds);
        if (i < nblk) totals[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *n_out = (int32_t)carry;
}
__global__ __launch_bounds__(COMPACT_BLOCK) void k_compact_scatter(const int32_t* __restrict__ rays_alive, uint32_t n,
                                                                    const uint32_t* __restrict__ local_prefix,
                                                                    const uint32_t* __restrict__ block_prefix,
                                                                    int32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * COMPACT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int32_t v = rays_alive[i];
    if (v >= 0) out[block_prefix[blockIdx.x] + local_prefix[i]] = v;
}


}  // namespace

// =====================================================================
// C ABI
// =====================================================================

extern "C" {

int lae_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                           float* nears, float* fars, void* stream) {
    if (N == 0) return LAE_OK;
    if (!rays_o || !rays_d || !aabb || !nears || !fars) return LAE_ENULL;
    k_near_far<<<lae::cdiv(N, 256), 256, 0, STREAM(stream)>>>(rays_o, rays_d, aabb, N, min_near, nears, fars);
    return lae::check_launch("near_far_from_aabb");
}

int lae_get_rays(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                 const int64_t* inds, uint64_t inds_batch_stride, uint32_t N, int perturb, float off_x, float off_y,
                 float* rays_o, float* rays_d, const float* aabb, float min_near, float* nears, float* fars, void* stream) {
    if (N == 0 || B == 0) return LAE_OK;
    if (!poses || !rays_o || !rays_d) return LAE_ENULL;
    if (aabb && (!nears || !fars)) return LAE_ENULL;
    if (H == 0 || W == 0 || B > 65535u || (!inds && (uint64_t)N != (uint64_t)H * W)) return LAE_EINVAL;
    k_get_rays<<<dim3(lae::cdiv(N, 256), B), 256, 0, STREAM(stream)>>>(poses, B, fx, fy, cx, cy, W, inds, inds_batch_stride, N, off_x,
                                                                       off_y, perturb, rays_o, rays_d, aabb, min_near, nears, fars);
    return lae::check_launch("get_rays");
}

int lae_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream) {
    if (N == 0) return LAE_OK;
    if (!rays_o || !rays_d || !coords) return LAE_ENULL;
    k_sph_from_ray<<<lae::cdiv(N, 256), 256, 0, STREAM(stream)>>>(rays_o, rays_d, radius, N, coords);
    return lae::check_launch("sph_from_ray");
}

int lae_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
    if (N == 0) return LAE_OK;
    if (!coords || !indices) return LAE_ENULL;
    k_morton3D<<<lae::cdiv(N, 256), 256, 0, STREAM(stream)>>>(coords, N, indices);
    return lae::check_launch("morton3D");
}

int lae_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
    if (N == 0) return LAE_OK;
    if (!coords || !indices) return LAE_ENULL;
    k_morton3D_invert<<<lae::cdiv(N, 256), 256, 0, STREAM(stream)>>>(indices, N, coords);
    return lae::check_launch("morton3D_invert");
}

int lae_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
    if (N == 0) return LAE_OK;
    if (!grid || !bitfield) return LAE_ENULL;
    k_packbits<<<lae::cdiv(N, 256), 256, 0, STREAM(stream)>>>(grid, N, density_thresh, bitfield);
    return lae::check_launch("packbits");
}

constexpr uint32_t MARCH_REC_RAYS_MAX = 1u << 18;         // replay records only for batches up to 256 k rays (73 MB of scratch)
static inline uint64_t march_base_bytes(uint32_t N) { return ((4ull * (2ull * N + 3) + 60) + 15) / 16 * 16; }
uint64_t lae_march_rays_train_scratch_bytes(uint32_t N) {
    // counts[N] | prefix[N + 2] | rows_end | (N <= 2^18:) nrec[N] | records[N * 24]
    return march_base_bytes(N) + (N <= MARCH_REC_RAYS_MAX ? ((4ull * N + 15) / 16 * 16) + (uint64_t)N * MARCH_REC_MAX * sizeof(MarchRec) : 0ull);
}

int lae_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                         const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                         const float* noises, void* scratch, uint32_t* rows_end_out, void* stream) {
    if (N == 0) return LAE_OK;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !xyzs || !dirs || !deltas || !rays || !noises || !scratch)
        return LAE_ENULL;
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0) return LAE_EINVAL;
    const MarchCfg cfg = make_cfg(bound, dt_gamma, max_steps, C, H);
    const uint32_t nblk = lae::cdiv(N, MARCH_WAVES);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* prefix = counts + N;
    hipStream_t s = STREAM(stream);
    MarchRecs recs{nullptr, nullptr};
    if (N <= MARCH_REC_RAYS_MAX) {
        uint8_t* base = reinterpret_cast<uint8_t*>(scratch) + march_base_bytes(N);
        recs.nrec = reinterpret_cast<uint32_t*>(base);
        recs.rec = reinterpret_cast<MarchRec*>(base + (4ull * N + 15) / 16 * 16);
    }
    k_march_train_wave<false><<<nblk, MARCH_BLOCK, 0, s>>>(rays_o, rays_d, grid, cfg, max_steps, N, M, nears, fars, noises,
                                                           counts, nullptr, nullptr, nullptr, nullptr, nullptr, recs);
    k_scan_counts<<<1, 1024, 0, s>>>(counts, N, M, prefix, counter, rows_end_out);
    k_march_train_wave<true><<<nblk, MARCH_BLOCK, 0, s>>>(rays_o, rays_d, grid, cfg, max_steps, N, M, nears, fars, noises,
                                                          counts, prefix, xyzs, dirs, deltas, rays, recs);
    return lae::check_launch("march_rays_train");
}

int lae_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                     uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth,
                                     float* image, void* stream) {
    if (N == 0) return LAE_OK;
    if (!rays || !weights_sum || !depth || !image) return LAE_ENULL;
    if (M > 0 && (!sigmas || !rgbs || !deltas)) return LAE_ENULL;
    k_composite_train_fwd<false><<<lae::cdiv(N, COMP_WAVES), COMP_BLOCK, 0, STREAM(stream)>>>(sigmas, rgbs, deltas, rays, M, N,
                                                                                               T_thresh, weights_sum, depth, image, Blend{});
    return lae::check_launch("composite_rays_train_forward");
}

int lae_composite_rays_train_forward_blend(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                           uint32_t M, uint32_t N, float T_thresh, const float* nears, const float* fars,
                                           const float* bg_rays, float bg_r, float bg_g, float bg_b, float* weights_sum,
                                           float* depth, float* image, float* depth_out, float* image_out, void* stream) {
    if (N == 0) return LAE_OK;
    if (!rays || !weights_sum || !depth || !image || !nears || !fars || !depth_out || !image_out) return LAE_ENULL;
    if (M > 0 && (!sigmas || !rgbs || !deltas)) return LAE_ENULL;
    const Blend bl{nears, fars, bg_rays, {bg_r, bg_g, bg_b}, image_out, depth_out};
    k_composite_train_fwd<true><<<lae::cdiv(N, COMP_WAVES), COMP_BLOCK, 0, STREAM(stream)>>>(sigmas, rgbs, deltas, rays, M, N,
                                                                                              T_thresh, weights_sum, depth, image, bl);
    return lae::check_launch("composite_rays_train_forward_blend");
}

int lae_composite_rays_train_step(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M,
                                  uint32_t N, float T_thresh, const float* nears, const float* fars, const float* bg_rays, float bg_r,
                                  float bg_g, float bg_b, const uint32_t* rows_end, const float* target, const float* scale,
                                  float* weights_sum, float* depth, float* image, float* depth_out, float* image_out,
                                  float* grad_image, float* grad_sigmas, float* grad_rgbs, float* loss_out, float* partials,
                                  int defer_loss, void* stream) {
    if (N == 0) return LAE_OK;
    if (!rays || !weights_sum || !depth || !image || !nears || !fars || !depth_out || !image_out || !rows_end || !target || !grad_image ||
        !loss_out || !partials)
        return LAE_ENULL;
    if (M > 0 && (!sigmas || !rgbs || !deltas || !grad_sigmas || !grad_rgbs)) return LAE_ENULL;
    const Blend bl{nears, fars, bg_rays, {bg_r, bg_g, bg_b}, image_out, depth_out};
    const StepLoss sl{target, scale, grad_image, partials, defer_loss ? loss_out : nullptr};
    const uint32_t nb = lae::cdiv(N, COMP_WAVES);
    k_composite_train_step<<<nb, COMP_BLOCK, 0, STREAM(stream)>>>(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image, bl,
                                                                 sl, rows_end, grad_sigmas, grad_rgbs);
    // defer_loss: the one-block sum of the partials (5.5 us + a kernel boundary on the step's critical path for a number that
    // feeds nothing on the device) is left to lae_loss_finish or to a later launch that takes it along
    // (lae_nerf_head_backward); loss_out holds NaN until then
    if (!defer_loss) k_loss_finish<<<1, 1024, 0, STREAM(stream)>>>(partials, nb, 3u * N, scale, loss_out);
    return lae::check_launch("composite_rays_train_step");
}

int lae_loss_finish(const float* partials, uint32_t n_part, uint32_t n_elem, const float* scale, float* loss_out, void* stream) {
    if (!partials || !loss_out) return LAE_ENULL;
    if (n_elem == 0) return LAE_EINVAL;
    k_loss_finish<<<1, 1024, 0, STREAM(stream)>>>(partials, n_part, n_elem, scale, loss_out);
    return lae::check_launch("loss_finish");
}

int lae_composite_rays_train_backward_blend_ex(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                               const float* rgbs, const float* deltas, const int32_t* rays,
                                               const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                               float T_thresh, const float* bg_rays, float bg_r, float bg_g, float bg_b,
                                               const uint32_t* rows_end, const float* grad_scale, float* grad_sigmas,
                                               float* grad_rgbs, void* stream) {
    if (N == 0 || M == 0) return LAE_OK;
    if (!grad_image || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image || !grad_sigmas || !grad_rgbs || !rows_end)
        return LAE_ENULL;                                   // grad_weights_sum may be NULL (= zero)
    const Dense dn{bg_rays, {bg_r, bg_g, bg_b}, rows_end, grad_scale};
    k_composite_train_bwd<true><<<lae::cdiv(N, COMP_WAVES), COMP_BLOCK, 0, STREAM(stream)>>>(
        grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs, dn);
    return lae::check_launch("composite_rays_train_backward_blend");
}

int lae_composite_rays_train_backward_blend(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                            const float* rgbs, const float* deltas, const int32_t* rays,
                                            const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                            float T_thresh, const float* bg_rays, float bg_r, float bg_g, float bg_b,
                                            const uint32_t* rows_end, float* grad_sigmas, float* grad_rgbs, void* stream) {
    if (N == 0 || M == 0) return LAE_OK;
    if (!grad_weights_sum) return LAE_ENULL;
    return lae_composite_rays_train_backward_blend_ex(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                                      T_thresh, bg_rays, bg_r, bg_g, bg_b, rows_end, nullptr, grad_sigmas, grad_rgbs, stream);
}

int lae_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                      const float* rgbs, const float* deltas, const int32_t* rays,
                                      const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                      float T_thresh, float* grad_sigmas, float* grad_rgbs, void* stream) {
    if (N == 0 || M == 0) return LAE_OK;
    if (!grad_weights_sum || !grad_image || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image ||
        !grad_sigmas || !grad_rgbs)
        return LAE_ENULL;
    k_composite_train_bwd<false><<<lae::cdiv(N, COMP_WAVES), COMP_BLOCK, 0, STREAM(stream)>>>(
        grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas,
        grad_rgbs, Dense{});
    return lae::check_launch("composite_rays_train_backward");
}

int lae_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                   float* dirs, float* deltas, const float* noises, void* stream) {
    (void)nears;
    if (n_alive == 0 || n_step == 0) return LAE_OK;
    if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !fars || !xyzs || !dirs || !deltas || !noises)
        return LAE_ENULL;
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0) return LAE_EINVAL;
    const MarchCfg cfg = make_cfg(bound, dt_gamma, max_steps, C, H);
    k_march_infer<false><<<lae::cdiv(n_alive, 128), 128, 0, STREAM(stream)>>>(
        n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, cfg, grid, nullptr, fars, xyzs, dirs, deltas, nullptr,
        noises);
    return lae::check_launch("march_rays");
}

int lae_march_rays_distill(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                           const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                           uint32_t C, uint32_t H, const uint8_t* grid, const uint8_t* edit_grid, const float* nears,
                           const float* fars, float* xyzs, float* dirs, float* deltas, uint8_t* edit_occ,
                           const float* noises, void* stream) {
    (void)nears;
    if (n_alive == 0 || n_step == 0) return LAE_OK;
    if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !edit_grid || !fars || !xyzs || !dirs || !deltas ||
        !edit_occ || !noises)
        return LAE_ENULL;
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0) return LAE_EINVAL;
    const MarchCfg cfg = make_cfg(bound, dt_gamma, max_steps, C, H);
    k_march_infer<true><<<lae::cdiv(n_alive, 128), 128, 0, STREAM(stream)>>>(
        n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, cfg, grid, edit_grid, fars, xyzs, dirs, deltas, edit_occ,
        noises);
    return lae::check_launch("march_rays_distill");
}

int lae_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                       const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                       float* image, void* stream) {
    if (n_alive == 0) return LAE_OK;
    if (!rays_alive || !rays_t || !sigmas || !rgbs || !deltas || !weights_sum || !depth || !image) return LAE_ENULL;
    k_composite_infer<false><<<lae::cdiv(n_alive, 128), 128, 0, STREAM(stream)>>>(
        n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, nullptr, depth, nullptr,
        nullptr, image);
    return lae::check_launch("composite_rays");
}

int lae_composite_rays_distill(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                               const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                               float* weights_edit_sum, float* depth, float* depth_edit, const uint8_t* edit_occ,
                               float* image, void* stream) {
    if (n_alive == 0) return LAE_OK;
    if (!rays_alive || !rays_t || !sigmas || !rgbs || !deltas || !weights_sum || !weights_edit_sum || !depth ||
        !depth_edit || !edit_occ || !image)
        return LAE_ENULL;
    k_composite_infer<true><<<lae::cdiv(n_alive, 128), 128, 0, STREAM(stream)>>>(
        n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, weights_edit_sum, depth,
        depth_edit, edit_occ, image);
    return lae::check_launch("composite_rays_distill");
}

uint64_t lae_compact_scratch_bytes(uint32_t n_alive) {
    return 4ull * ((uint64_t)n_alive + lae::cdiv(n_alive, COMPACT_BLOCK)) + 64;
}

int lae_compact_rays_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out_alive, int32_t* n_out_dev,
                           void* scratch, void* stream) {
    if (!n_out_dev) return LAE_ENULL;
    hipStream_t s = STREAM(stream);
    if (n_alive == 0) return hipMemsetAsync(n_out_dev, 0, 4, s) == hipSuccess ? LAE_OK : LAE_ELAUNCH;
    if (!rays_alive || !out_alive || !scratch) return LAE_ENULL;
    const uint32_t nblk = lae::cdiv(n_alive, COMPACT_BLOCK);
    uint32_t* local_prefix = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* totals = local_prefix + n_alive;
    k_compact_count<<<nblk, COMPACT_BLOCK, 0, s>>>(rays_alive, n_alive, local_prefix, totals);
    k_compact_scan<<<1, 1024, 0, s>>>(totals, nblk, n_out_dev);
    k_compact_scatter<<<nblk, COMPACT_BLOCK, 0, s>>>(rays_alive, n_alive, local_prefix, totals, out_alive);
    return lae::check_launch("compact_rays_alive");
}


}  // extern "C"
