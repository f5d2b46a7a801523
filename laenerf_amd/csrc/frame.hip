// frame.hip -- the device-resident whole-frame inference loop (`lae_render_frame`) for gfx950: the host loop of
// NeRFRenderer.run_cuda (nerf/renderer.py:335-387) and run_cuda_distill (:394-480) as ONE backend call with its loop state on
// the device -- lookahead marcher on a side stream, poll / poison hand-over, admission control, emit kernels, degrade path.
// Split out of raymarching.hip in round 6 (the operators stay there); shares the marcher geometry through raymarch_common.h.
// Same arithmetic contract as raymarching.hip (-ffp-contract=off, explicit fmaf()).
#include <string.h>
#include <algorithm>
#include <chrono>
#include <climits>
#include <mutex>
#include "lae_common.h"
#include "raymarch_common.h"

namespace {

// ---------------------------------------------------------------- whole-frame inference loop (MI355X-native)
// The reference renders a frame with a HOST loop (renderer.py:352-379): every iteration launches march_rays, the
// network, composite_rays, compacts the alive list with a boolean mask and reads its length back -- one device sync and
// ~20 launches per iteration, ~75 iterations per 800x800 frame.  Here the loop state lives in device memory:
//   * k_frame_composite: each workgroup owns a contiguous range of the alive list and writes its survivors, in order,
//     into its own segment + a per-segment count (no atomics, deterministic order);
//   * k_frame_emit of the NEXT iteration: every workgroup sums the <= 512 segment counts (n_alive and its own segment's
//     offset), derives the loop state exactly like the Python (`n_step = max(min(N // n_alive, 8), 1)`, `step += n_step`,
//     stop when no ray is alive or step >= max_steps), gathers its rays into the compact alive list and writes their sample rows;
//     workgroup 0 publishes the state (FrameCtrl, double-buffered) for the encoder / MLP / compositing kernels and
//     mirrors n_alive into pinned host memory, from which the host sizes later launches without ever synchronising.
// Marching is split in two, because sample positions are pure geometry (they do not depend on the network):
//   * k_frame_lookahead walks every alive ray ahead and records the times of its next <= 8 samples (max_n_step); it runs
//     on a SIDE STREAM concurrently with the encoder / MLP kernels of the same iteration.  A lane walks its ray like the
//     reference for a few visits; lanes still crossing empty space afterwards (~200 instructions per voxel at 1/64
//     utilisation; measured: every iteration waited ~100 us for such stragglers when this ran in-line) are finished one
//     by one with all 64 lanes when they are few (frame_lookahead_coop, the candidate scheme of the training march).
//   * k_frame_emit (in-line) replays the first n_step recorded samples of every surviving ray into the sample rows:
//     positions and deltas are recomputed from the recorded times with the reference's arithmetic, no probing.
// The lookahead also advances its own copy of rays_t with the compositing kernel's arithmetic (t += deltas[1] per
// sample), so every iteration starts from bit-identical times to the host loop's.
using lae::FrameCtrl;                          // lae_common.h (shared with the head kernel, ffmlp.hip)
using lae::RayAcc;
struct FrameMirror { volatile uint64_t tag; volatile uint32_t n_alive, done, total_rows, iters, hung; };   // pinned host memory; hung: low 32 bits of the frame number in which a k_frame_wait gave up (0: never)
constexpr int FRAME_BLOCK = 256;
constexpr uint32_t FRAME_SEG_MAX = 4096;       // survivor segments = waves of the head + compositing kernel (k_frame_head, ffmlp.hip)
constexpr uint32_t FRAME_SEG_SLACK = 200;      // a segment's stride exceeds bound_alive / segments by < 66 + 63 + 64 (see the stride below)
constexpr uint32_t FRAME_LA = 8;               // recorded samples per ray (>= max_n_step)

__global__ void k_frame_init(FrameCtrl* __restrict__ ctrl, uint32_t N, const float* __restrict__ nears,
                             RayAcc* __restrict__ acc, float* __restrict__ tc, uint32_t* __restrict__ q_counts, uint32_t* __restrict__ cmask) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (cmask && blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < 32u * 32u; i += blockDim.x) cmask[i] = 0u;   // the coarse mask of k_frame_coarse_mark (FRAME_CG^2 words)
    if (n == 0) { ctrl[0] = FrameCtrl{}; q_counts[0] = 0u; q_counts[1] = 0u; q_counts[2] = 0u; q_counts[3] = 0u; }   // state "before iteration 0": step = 0, nothing issued, no straggler queued
    if (n >= N) return;
    const float t0 = nears[n];
    tc[n] = t0;
    reinterpret_cast<float4*>(acc + n)[0] = make_float4(0.f, 0.f, 0.f, 0.f);      // weights_sum, depth, image
    reinterpret_cast<float4*>(acc + n)[1] = make_float4(0.f, t0, 0.f, 0.f);       // image.b, rays_t, weights_edit_sum, depth_edit
}

// ---- "nothing ahead" test of the lookahead.  The slowest waves of a lookahead launch are those whose rays have left the
// surface and cross empty space to `far`: 16-25 rounds of 8 visits (130-200 us) while the median wave lives 11 us -- the
// launch's span is that tail (profiles/r3b_frame_look_stamps.txt).  A walk that will find nothing may simply stop: the reference's
// walker would visit every cell up to `far`, emit no sample and leave the ray with fewer than n_step samples, which is all the
// compositing kernel looks at (raymarching.cu:881, 926).  So once per frame the bitfield is reduced to a 32^3 world-space
// grid of "a sample positioned here could probe an occupied cell of SOME cascade level" (every occupied 2x2x2 block -- one
// bitfield byte -- dilated by one cell of its level, which covers any rounding in the reference's index arithmetic), and that
// to a Chebyshev distance field; a lane whose ray is still unfinished after a round sphere-traces the field from t to `far`
// and, if every point of the rest of the ray keeps at least one coarse cell between itself and any marked cell, is done.
// Conservative: a failed test changes nothing (the lane walks on), a passed one only skips visits that emit nothing.
constexpr uint32_t FRAME_CG = 32;              // coarse cells per axis
constexpr uint32_t FRAME_CG_CAP = 8;           // distances saturate here (a step of the trace is at most 7 cells; 16: twice the rounds of the one-workgroup transform, 22 us)
// one thread per 4 bitfield bytes (4 Morton-consecutive 2x2x2 blocks); a workgroup first collects its marks in an LDS copy of the
// mask and flushes the words it set (one global atomic per word and workgroup: per byte, the atomics of a scene with a wall
// across the volume took 65 us)
__global__ __launch_bounds__(1024) void k_frame_coarse_mark(const uint8_t* __restrict__ grid, uint32_t C, uint32_t H, float bound, uint32_t* __restrict__ cmask) {
    __shared__ uint32_t lm[FRAME_CG * FRAME_CG];
    for (uint32_t i = threadIdx.x; i < FRAME_CG * FRAME_CG; i += blockDim.x) lm[i] = 0u;
    __syncthreads();
    const uint32_t per_level = H * H * H / 8u, total = C * per_level;
    const float inv_s = (float)FRAME_CG / (2.0f * bound);
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w * 4u < total; w += gridDim.x * blockDim.x) {
        const uint32_t word = reinterpret_cast<const uint32_t*>(grid)[w];       // (H >= 4: per_level is a multiple of 4, the bitfield 4-byte aligned)
        if (word == 0u) continue;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            if (((word >> (8u * k)) & 0xffu) == 0u) continue;
            const uint32_t b = w * 4u + k;
            const uint32_t level = b / per_level, m = (b - level * per_level) << 3;
            const uint32_t v[3] = {morton_compact(m), morton_compact(m >> 1), morton_compact(m >> 2)};   // even: the block's low corner
            const float bl = fminf(scalbnf(1.0f, (int)level), bound), vs = 2.0f * bl / (float)H;
            const bool top = level + 1u == C;             // positions beyond the top level's box clamp into its border cells (probe_at)
            int c0[3], c1[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                float lo = -bl + ((float)v[a] - 1.0f) * vs, hi = -bl + ((float)v[a] + 3.0f) * vs;   // cells v-1 .. v+2
                if (top && v[a] == 0u) lo = -bound;
                if (top && v[a] + 2u >= H) hi = bound;
                c0[a] = max(0, min((int)FRAME_CG - 1, (int)floorf((lo + bound) * inv_s)));
                c1[a] = max(0, min((int)FRAME_CG - 1, (int)floorf((hi + bound) * inv_s)));
            }
            const uint32_t xm = (c1[0] >= 31 ? ~0u : ((2u << c1[0]) - 1u)) & ~((1u << c0[0]) - 1u);
            for (int z = c0[2]; z <= c1[2]; z++)
                for (int y = c0[1]; y <= c1[1]; y++)
                    if ((lm[z * (int)FRAME_CG + y] & xm) != xm) atomicOr(&lm[z * (int)FRAME_CG + y], xm);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < FRAME_CG * FRAME_CG; i += blockDim.x) {
        const uint32_t mine = lm[i];
        if (mine && (__hip_atomic_load(cmask + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mine) != mine) atomicOr(cmask + i, mine);
    }
}
// Chebyshev distance (in coarse cells, saturating at FRAME_CG_CAP) to the nearest marked cell: rounds of 3x3x3 dilation on the
// 1024 x-rows of the mask held as 32-bit words; one workgroup.
__device__ __forceinline__ void frame_coarse_octants(const uint32_t* __restrict__ cmask, uint8_t* __restrict__ coct, uint32_t (*rows)[1024]);
__global__ __launch_bounds__(1024) void k_frame_coarse_dist(const uint32_t* __restrict__ cmask, uint8_t* __restrict__ cdist) {
    static_assert(FRAME_CG == 32, "one 32-bit word per x-row");
    __shared__ uint32_t rows[2][1024];
    if (blockIdx.x == 1) { frame_coarse_octants(cmask, cdist + FRAME_CG * FRAME_CG * FRAME_CG, rows); return; }   // the second workgroup: the octant flags
    const uint32_t tid = threadIdx.x, y = tid & 31u, z = tid >> 5;
    uint32_t cur = cmask[tid];
    for (uint32_t x = 0; x < 32u; x++) cdist[tid * 32u + x] = ((cur >> x) & 1u) ? 0 : (uint8_t)FRAME_CG_CAP;
    for (uint32_t r = 1; r < FRAME_CG_CAP; r++) {
        rows[0][tid] = cur | (cur << 1) | (cur >> 1);
        __syncthreads();
        uint32_t a = rows[0][tid];
        if (y > 0) a |= rows[0][tid - 1];
        if (y < 31) a |= rows[0][tid + 1];
        rows[1][tid] = a;
        __syncthreads();
        uint32_t d = rows[1][tid];
        if (z > 0) d |= rows[1][tid - 32];
        if (z < 31) d |= rows[1][tid + 32];
        uint32_t fresh = d & ~cur;
        while (fresh) { const uint32_t x = (uint32_t)__builtin_ctz(fresh); fresh &= fresh - 1u; cdist[tid * 32u + x] = (uint8_t)r; }
        cur = d;
        __syncthreads();
    }
}
// Per coarse cell and direction octant (bit o = (dx < 0) | (dy < 0) << 1 | (dz < 0) << 2): is the whole sub-box from this cell to
// the octant's corner of the volume unmarked?  A ray's coordinates are monotone in t, so from a point in this cell it never
// leaves that sub-box: ONE byte load answers "nothing ahead" for the common case of a ray that has passed the marked region's
// extent on some axis (the sphere trace below answers the rest, at 5-10 dependent loads).  Workgroup 1 of k_frame_coarse_dist.
__device__ __forceinline__ void frame_coarse_octants(const uint32_t* __restrict__ cmask, uint8_t* __restrict__ coct, uint32_t (*rows)[1024]) {
    uint32_t* a = rows[0]; uint32_t* b = rows[1];
    const uint32_t tid = threadIdx.x, y = tid & 31u, z = tid >> 5;
    const uint32_t w = cmask[tid];
    uint32_t clear[8];
#pragma unroll
    for (uint32_t o = 0; o < 8u; o++) {
        uint32_t s = w;                                    // bit x: a marked cell at x' >= x (dx >= 0) / x' <= x (dx < 0) of this row
        if (o & 1u) { s |= s << 1; s |= s << 2; s |= s << 4; s |= s << 8; s |= s << 16; }
        else { s |= s >> 1; s |= s >> 2; s |= s >> 4; s |= s >> 8; s |= s >> 16; }
        __syncthreads();
        a[tid] = s;
        __syncthreads();
        uint32_t t = 0;
        if (o & 2u) { for (uint32_t yy = 0; yy <= y; yy++) t |= a[z * 32u + yy]; }
        else { for (uint32_t yy = y; yy < 32u; yy++) t |= a[z * 32u + yy]; }
        b[tid] = t;
        __syncthreads();
        uint32_t u = 0;
        if (o & 4u) { for (uint32_t zz = 0; zz <= z; zz++) u |= b[zz * 32u + y]; }
        else { for (uint32_t zz = z; zz < 32u; zz++) u |= b[zz * 32u + y]; }
        clear[o] = ~u;
    }
    for (uint32_t x = 0; x < 32u; x++) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t o = 0; o < 8u; o++) v |= ((clear[o] >> x) & 1u) << o;
        coct[tid * 32u + x] = (uint8_t)v;
    }
}
__device__ __forceinline__ uint32_t frame_coarse_cell(const Ray& r, float bound, float t) {
    const float inv_s = (float)FRAME_CG / (2.0f * bound);
    const int cx = max(0, min((int)FRAME_CG - 1, (int)floorf((fmaf(t, r.dx, r.ox) + bound) * inv_s)));
    const int cy = max(0, min((int)FRAME_CG - 1, (int)floorf((fmaf(t, r.dy, r.oy) + bound) * inv_s)));
    const int cz = max(0, min((int)FRAME_CG - 1, (int)floorf((fmaf(t, r.dz, r.oz) + bound) * inv_s)));
    return (uint32_t)((cz * (int)FRAME_CG + cy) * (int)FRAME_CG + cx);
}
// the octant test alone (coct = cdist + FRAME_CG^3)
__device__ __forceinline__ bool frame_clear_octant(const Ray& r, float bound, float t, const uint8_t* __restrict__ cdist) {
    const uint32_t o = (r.dx < 0.0f ? 1u : 0u) | (r.dy < 0.0f ? 2u : 0u) | (r.dz < 0.0f ? 4u : 0u);
    return (cdist[FRAME_CG * FRAME_CG * FRAME_CG + frame_coarse_cell(r, bound, t)] >> o) & 1u;
}
// true: no cell the rest of the ray [t, far) could probe is occupied
__device__ __forceinline__ bool frame_clear_to_far(const Ray& r, float bound, float t, float far, const uint8_t* __restrict__ cdist) {
    // a step of (D - 1) coarse cells in the largest direction component stays inside the cube of cells D - 1 rings around the
    // current one, all of them unmarked; 0.98: rounding of the products below
    const float per_cell = 0.98f * (2.0f * bound / (float)FRAME_CG) / fmaxf(fabsf(r.dx), fmaxf(fabsf(r.dy), fabsf(r.dz)));
    const uint32_t o = (r.dx < 0.0f ? 1u : 0u) | (r.dy < 0.0f ? 2u : 0u) | (r.dz < 0.0f ? 4u : 0u);
    for (int it = 0; it < 24; it++) {
        if (!(t < far)) return true;
        const uint32_t cell = frame_coarse_cell(r, bound, t);
        const uint32_t D = cdist[cell];
        if ((cdist[FRAME_CG * FRAME_CG * FRAME_CG + cell] >> o) & 1u) return true;
        if (D < 2u) return false;
        t = fmaf((float)(D - 1u), per_cell, t);
    }
    return false;
}

// Wave-cooperative continuation of ONE ray (state broadcast from lane L): the candidate scheme of k_march_train_wave --
// 64 consecutive visit candidates T_k per pass, probed in parallel, visited chain resolved from ballots -- recording up
// to `remaining` sample times at out_t[0..) (+ edit flags).  Exactly the serial walk's samples (same candidate times,
// same probe arithmetic); ~14 voxels of empty space per pass instead of one visit per ~200 instructions of one lane.
// Returns the number of samples recorded (wave-uniform).
template <bool EDIT>
__device__ __forceinline__ uint32_t frame_lookahead_coop(const Ray& r, const MarchCfg& cfg, const uint8_t* __restrict__ grid,
                                                         const uint8_t* __restrict__ edit_grid, float t_base, float far,
                                                         uint32_t remaining, float* __restrict__ out_t, uint8_t* __restrict__ out_e,
                                                         int lane, float& t_end, const uint8_t* __restrict__ cdist = nullptr) {
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t emitted = 0;
    bool pending = false;
    float pending_tt = 0.f;
    uint32_t idle = 0;                                     // passes since the last sample
    t_end = t_base;                                        // the serial walker's t when it stops (wave-uniform)
    while (t_base < far && emitted < remaining) {
        // every fourth pass without a sample: is anything ahead at all?  (wave-uniform; k_frame_coarse_mark)
        if (cdist && (idle & 3u) == 3u && frame_clear_to_far(r, cfg.bound, t_base, far, cdist)) { t_end = far; break; }
        idle++;
        const float t = candidate_t(cfg, t_base, lane);
        const float t_next = t + step_of(cfg, t);
        const bool valid = t < far;
        Probe p;
        p.occ = false; p.tt = t; p.x = p.y = p.z = 0.f; p.dt = 0.f; p.index = 0;
        if (valid) p = probe_at(r, cfg, grid, t);
        const unsigned long long valid_mask = __ballot(valid);
        const unsigned long long occ_mask = __ballot(valid && p.occ);
        unsigned long long emit = 0ull;
        int k = 0;
        if (pending) {
            const unsigned long long m = __ballot(valid && t >= pending_tt);
            if (m) { k = __builtin_ctzll(m); pending = false; } else k = 64;
        }
        while (k < 64 && ((valid_mask >> k) & 1ull)) {
            if ((occ_mask >> k) & 1ull) {
                const unsigned long long inv = (~occ_mask) >> k;
                const int run = inv ? __builtin_ctzll(inv) : 64 - k;
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
        uint32_t cnt = (uint32_t)__builtin_popcountll(emit);
        bool done = false;
        if (emitted + cnt > remaining) {
            const uint32_t keep = remaining - emitted;
            const bool mine = ((emit >> lane) & 1ull) && (uint32_t)__builtin_popcountll(emit & below) < keep;
            emit = __ballot(mine);
            cnt = keep;
            done = true;
        }
        if ((emit >> lane) & 1ull) {
            const uint32_t slot = emitted + (uint32_t)__builtin_popcountll(emit & below);
            out_t[slot] = t;
            if (EDIT) out_e[slot] = (edit_grid[p.index >> 3] >> (p.index & 7u)) & 1u;
        }
        emitted += cnt;
        if (cnt) idle = 0;
        if (emit && emitted >= remaining) {                // sample budget reached: the walker stands right after its last sample
            t_end = __shfl(t_next, 63 - __builtin_clzll(emit), 64);
            break;
        }
        if (done || valid_mask != ~0ull) { t_end = far; break; }   // the ray left [near, far): nothing further to find
        t_base = __shfl(t_next, 63, 64);
        t_end = t_base;
    }
    return emitted;
}

__device__ __forceinline__ float bcast(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// start of the walk in the first iteration: t = near + clamp(near * dt_gamma) * noise (raymarching.cu:746)
__device__ __forceinline__ float perturbed_start(const MarchCfg& cfg, float t, const float* __restrict__ noises, uint32_t n) {
    return noises ? fmaf(clampf(t * cfg.dt_gamma, cfg.dt_min, cfg.dt_max), noises[n], t) : t;
}

// Lookahead marcher.  phase -1: before the loop (walk from the perturbed near).  phase p >= 0: after k_frame_emit of
// iteration p -- first advance tc[ray] (this loop's copy of rays_t) over the n_step samples that iteration consumes, with
// the compositing kernel's arithmetic (t += deltas[1]), then walk on from there (march_rays restarts from rays_t,
// raymarching.cu:736).  Records la_t[ray][0..cnt) = sample times, la_e = edit flags, la_cnt[ray] = cnt <= max_n_step.
// ---- survivors of the previous iteration: nu segments (one per wave of k_frame_head, in list order) of stride R with their
// counts, and the per-workgroup sums of those counts (FRAME_HEAD_WAVES consecutive segments = one workgroup).  Both per-ray
// kernels of an iteration (emit, lookahead) walk these lists directly: wave wv of a launch takes entries
// [(wv % cpb) * 64, +64) of workgroup-level list wv / cpb, cpb = W * R / 64, finds its lane's segment from the workgroup's W
// counts, and gets n_alive and its offset in the compact order from the <= 512 workgroup sums (the per-wave counts of every
// workgroup, 2048 words read by each of ~14 k waves, made the emit kernel 18 us instead of 13).
struct FrameSegs { const uint32_t* unit_counts; const uint32_t* blk_counts; const int32_t* seg; uint32_t nu, R; };
struct FrameSlot { uint32_t n_alive, n, index; bool has; };
__device__ __forceinline__ FrameSlot frame_locate(const FrameSegs& sg, uint32_t N, uint32_t wv, int lane) {
    constexpr uint32_t W = lae::FRAME_HEAD_WAVES;
    FrameSlot f;
    if (sg.nu == 0) {                                      // first iteration: every ray, identity order
        f.n_alive = N; f.n = wv * 64u + (uint32_t)lane; f.index = f.n; f.has = f.n < N;
        return f;
    }
    const uint32_t nb = sg.nu / W, cpb = W * sg.R / 64u;
    const uint32_t bb = wv / cpb, l = (wv % cpb) * 64u + (uint32_t)lane;
    const uint32_t bbc = bb < nb ? bb : 0u;
    uint32_t below = 0, total = 0;
    for (uint32_t j = (uint32_t)lane; j < nb; j += 64u) {
        const uint32_t c = sg.blk_counts[j];
        total += c; below += j < bb ? c : 0u;
    }
    uint32_t uc[W];                                        // requested beside the sums above: one round trip, not two
#pragma unroll
    for (uint32_t j = 0; j < W; j++) uc[j] = sg.unit_counts[bbc * W + j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { total += __shfl_xor(total, d, 64); below += __shfl_xor(below, d, 64); }
    uint32_t pre = 0, unit = bbc * W, off = l;
    bool found = false;
#pragma unroll
    for (uint32_t j = 0; j < W; j++) {
        if (!found && l < pre + uc[j]) { found = true; unit = bbc * W + j; off = l - pre; }
        pre += uc[j];
    }
    f.n_alive = total; f.n = below + l; f.index = 0;
    f.has = bb < nb && found;                              // pre == the workgroup's survivors
    if (f.has) f.index = (uint32_t)sg.seg[(size_t)unit * sg.R + off];
    return f;
}
// Cross-stream handshake of the frame loop WITHOUT events: hipEventRecord + hipStreamWaitEvent cost 10-12 us per dependency
// on this stack, a 64-bit word in device memory written by the producer (its own kernel, or a one-thread kernel behind it)
// and a one-wave kernel polling it on the consumer's stream 1.7-2.8 us (tools/ubench/stream_hop.hip, profiles/r4_stream_hop.txt).
// Values grow monotonically over the life of the process ((frame number << 32) | iteration + 1), so nothing is ever reset.
__global__ void k_frame_signal(unsigned long long* __restrict__ flag, unsigned long long value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// A wait that gives up POISONS its frame (round 5): it stores the frame's tag in a device word the lookahead
// kernels test before they touch anything (the segments they would read are not there yet) and k_frame_finish tests before it
// writes the outputs (NaN instead of an image composited from rows that were never marched), and in the pinned mirror for the
// host, which then renders the frame again with the lookahead in line (lae_render_frame).  Later waits of a poisoned frame
// return at once, so an aborted frame drains in one time-out, not one per queued wait.
__global__ void k_frame_wait(const unsigned long long* __restrict__ flag, unsigned long long value, uint32_t* __restrict__ hung_dev,
                             volatile uint32_t* __restrict__ hung_host, uint32_t tag, uint32_t spins) {
    if (__hip_atomic_load(hung_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) return;
    // bounded (2^23 polls ~ 10 s): a producer that never comes (a failed launch on the other stream, two streams that share one
    // hardware queue) must not hang the device
    for (uint32_t spin = 0; spin < spins; spin++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= value) return;
        __builtin_amdgcn_s_sleep(4);
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(hung_dev, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        *hung_host = tag;
    }
}
// one-time probe per caller stream (lae_render_frame): can a kernel on `a` see a store made by a kernel that was launched LATER
// on `b`?  result: 1 = yes (the streams run side by side), 2 = no (one hardware queue, serialised dispatch, a profiler that
// collects counters): the frame loop then runs its lookahead in line
__global__ void k_frame_probe_wait(const unsigned long long* __restrict__ flag, unsigned long long value, volatile uint32_t* __restrict__ result,
                                   uint32_t spins) {
    for (uint32_t spin = 0; spin < spins; spin++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= value) { if (threadIdx.x == 0) *result = 1u; return; }
        __builtin_amdgcn_s_sleep(4);
    }
    if (threadIdx.x == 0) *result = 2u;
}
// the next loop state from the previous one and the number of survivors (renderer.py:352,363,377); evaluated with the same
// arguments by the emit kernel (which publishes it) and by the lookahead kernel (which only needs n_step and done)
__device__ __forceinline__ FrameCtrl frame_next_ctrl(const FrameCtrl& pv, uint32_t n_alive, uint32_t row_budget, uint32_t max_steps,
                                                     uint32_t max_n_step) {
    FrameCtrl c;
    c.step = pv.step + pv.n_step;
    c.total_rows = pv.total_rows + pv.n_alive * pv.n_step;
    c.done = (pv.done || n_alive == 0 || c.step >= max_steps) ? 1u : 0u;
    c.n_alive = c.done ? 0u : n_alive;
    c.n_step = c.done ? 0u : max(min(row_budget / n_alive, max_n_step), 1u);
    const uint32_t rpg = lae::frame_rays_per_group(max(c.n_step, 1u));          // rows in 64-row groups of whole rays (lae_common.h)
    const uint32_t n_groups = (c.n_alive + rpg - 1u) / rpg;
    c.n_rows = c.n_alive ? (n_groups - 1u) * 64u + (c.n_alive - (n_groups - 1u) * rpg) * c.n_step : 0u;
    c.iter = pv.iter + (c.done ? 0u : 1u);
    c.pad = 0;
    return c;
}
// a ray's lookahead record, double-buffered: iteration i's emit AND lookahead read buffer i & 1, the lookahead writes the other
struct LookRec { float* t; uint8_t* e; uint32_t* cnt; float* tend; float* tc; };      // [N * FRAME_LA], [N * FRAME_LA], [N], [N], [N]

// in-kernel stamps of the lookahead for tools/frame_look_stamps.py (compiled in only with -DLAE_FRAME_STAMPS): per wave of ONE
// launch (phase == g_look_phase): wall clock at start / state loaded / lane rounds done / end, lane rounds, coop rays, coop passes
#ifdef LAE_FRAME_STAMPS
__device__ unsigned long long g_look_stamps[16384 * 8];
__device__ int g_look_phase = 20;
#define LOOK_NOTE(i, v) do { if (phase == g_look_phase && lane == 0) { const uint32_t w_ = blockIdx.x * (FRAME_BLOCK / 64) + (threadIdx.x >> 6); if (w_ < 16384u) g_look_stamps[(size_t)w_ * 8 + (i)] = (v); } } while (0)
#else
#define LOOK_NOTE(i, v) do { } while (0)
#endif
constexpr uint32_t FRAME_LANE_VISITS = 8;      // visits a lane walks alone before the wave decides how to continue
constexpr int FRAME_QUEUE_MAX = 16;            // the bound when the stragglers go to the finishing kernel (8 / 16 / 24 / 32 / 48: 12.07 / 11.86 / 11.88 / 11.98 / 12.14 ms per frame)
constexpr int FRAME_MAX_ROUNDS = 1 << 30;      // lane rounds after which ALL unfinished lanes of a wave go to the finishing kernel: never (2: 800x800 11.9 -> 15.1 ms, 1080p 71 -> 92 ms)
constexpr int FRAME_SPEC = 8;                  // visits of a walk through empty space laid out (and probed) together
struct LookTask { uint32_t index, step; float t; };   // a ray the lane phase hands to k_frame_lookahead_finish
constexpr uint32_t FRAME_FINISH_BLOCKS = 1024; // x 4 waves: more than one wave per SIMD, tasks dealt round-robin
constexpr int FRAME_COOP_MAX = 8;              // unfinished lanes per wave up to which they are finished cooperatively
template <bool EDIT>
__global__ __launch_bounds__(FRAME_BLOCK) void k_frame_lookahead(
    int phase, const FrameCtrl* __restrict__ prev, FrameSegs sg, uint32_t N, uint32_t row_budget, uint32_t max_steps,
    uint32_t max_n_step, LookRec in, LookRec out,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ fars, MarchCfg cfg,
    const uint8_t* __restrict__ grid, const uint8_t* __restrict__ edit_grid, const float* __restrict__ noises,
    uint32_t* __restrict__ q_count, LookTask* __restrict__ q_tasks, int spec, int coop_max, int max_rounds,
    const uint8_t* __restrict__ cdist, uint32_t admit_cap, int admit_round, const uint32_t* __restrict__ hung_dev, uint32_t hung_tag) {
    if (*hung_dev == hung_tag) return;                     // a wait of this frame gave up: the survivor segments are not there (k_frame_wait)
    const int lane = threadIdx.x & 63;
    const uint32_t wv = blockIdx.x * (FRAME_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    LOOK_NOTE(0, wall_clock64());
    // Iteration `phase` consumes n_step samples of every ray that survived iteration phase - 1.  Neither the rays nor n_step
    // come from this iteration's emit kernel: the rays are read off the previous head kernel's survivor segments, n_step
    // follows from their number (frame_next_ctrl, the emit kernel's own formula), and the records are double-buffered -- so
    // this launch needs nothing the emit kernel of its iteration writes and may run beside it.
    FrameSlot f;
    uint32_t n_consumed = 0;
    if (phase < 0) { f.n_alive = N; f.n = wv * 64u + (uint32_t)lane; f.index = f.n; f.has = f.n < N; }
    else {
        f = frame_locate(sg, N, wv, lane);
        const FrameCtrl c = frame_next_ctrl(*prev, f.n_alive, row_budget, max_steps, max_n_step);
        if (c.done) return;
        n_consumed = c.n_step;
    }
    const uint32_t n = f.n;
    uint32_t index = f.index;
    bool has_ray = f.has;
    if (__ballot(has_ray) == 0ull) return;
    if (!has_ray) index = 0;
    uint32_t st_rounds = 0;
    bool admit_tried = false;
    [[maybe_unused]] uint32_t st_coop = 0, st_passes = 0;
    Ray r{};
    float t = 0.f, far = 0.f, tend0 = 0.f;
    uint32_t cnt0 = 0;
    float* out_t = out.t + (size_t)index * FRAME_LA;
    uint8_t* out_e = EDIT ? out.e + (size_t)index * FRAME_LA : nullptr;
    // everything the ray's index addresses is requested together, before the count decides whether the ray goes on; the
    // record travels as two 16-byte loads (a load per consumed sample and a load + store per kept one, each waited for in
    // turn, were a dozen dependent round trips through a memory system the encoder kernel next door keeps saturated)
    static_assert(FRAME_LA == 8, "the record is handled as 2 x float4 / one 64-bit word");
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    unsigned long long rece = 0ull;
    if (has_ray) {
        r = load_ray(rays_o, rays_d, index);
        far = fars[index];
        t = in.tc[index];
        if (phase >= 0) {
            cnt0 = in.cnt[index];
            ra = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[0]; rb = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[1];
            if (EDIT) rece = *reinterpret_cast<const unsigned long long*>(in.e + (size_t)index * FRAME_LA);
            tend0 = in.tend[index];
        }
    }
    if (phase >= 0 && cnt0 < n_consumed) has_ray = false;  // the ray ends in this iteration
    if (__ballot(has_ray) == 0ull) return;                 // whole wave idle; otherwise ray-less lanes stay as helpers
    uint32_t step = 0;
    if (has_ray) {
        if (phase < 0) t = perturbed_start(cfg, t, noises, n);
        else {
            float rec[FRAME_LA] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
            float last = phase == 0 ? perturbed_start(cfg, t, noises, n) : t;   // iteration 0 lists rays in identity order
#pragma unroll
            for (uint32_t j = 0; j < FRAME_LA; j++) {
                if (j < n_consumed) {                      // n_consumed is uniform over the launch
                    const float tj = rec[j];
                    const float tn = tj + step_of(cfg, tj);
                    t += tn - last;                        // composite: t += deltas[1], deltas[1] = t_next - last_t
                    last = tn;
                }
            }
            out.tc[index] = t;
            // The samples recorded beyond the consumed ones are exactly what a walk restarted at t would find, provided the
            // compositing kernel's running t equals the walker's own t after the last consumed sample (it does unless the
            // float subtraction / addition pair above rounded: then everything is walked again from t, like the reference).
            if (n_consumed > 0 && t == last) {
                const uint32_t keep = cnt0 - n_consumed;
                if (keep) {                                // shift by the uniform n_consumed: three conditional register moves
                    if (n_consumed & 1u) {
#pragma unroll
                        for (int j = 0; j < 7; j++) rec[j] = rec[j + 1];
                    }
                    if (n_consumed & 2u) {
#pragma unroll
                        for (int j = 0; j < 6; j++) rec[j] = rec[j + 2];
                    }
                    if (n_consumed & 4u) {
#pragma unroll
                        for (int j = 0; j < 4; j++) rec[j] = rec[j + 4];
                    }
                    reinterpret_cast<float4*>(out_t)[0] = make_float4(rec[0], rec[1], rec[2], rec[3]);
                    reinterpret_cast<float4*>(out_t)[1] = make_float4(rec[4], rec[5], rec[6], rec[7]);
                    if (EDIT) *reinterpret_cast<unsigned long long*>(out_e) = rece >> (8u * n_consumed);
                }
                step = keep;
                t = tend0;
            }
        }
    }
    // the first walk of a frame: two thirds of an 800x800 lego frame's rays cross the volume without touching anything
    if (cdist && phase < 0 && has_ray && t < far && frame_clear_to_far(r, cfg.bound, t, far, cdist)) t = far;
    LOOK_NOTE(1, wall_clock64());
    for (;;) {
        st_rounds++;
        uint32_t visits = 0;                               // lane phase: the reference's walk
        const uint32_t step_before = step;
        // A lane crossing empty space pays one DEPENDENT bitfield probe per visit, and a wave with more than FRAME_COOP_MAX
        // such lanes keeps walking (8-13 rounds of 8 visits: the slowest waves once the few-straggler case went to the
        // finishing kernel).  The visits of an empty stretch do not depend on what the probes return as long as they
        // return "empty": after a first ordinary visit that found nothing, a lane lays out the next FRAME_SPEC visits
        // under that assumption, requests their probes together and takes them in order; the first occupied one is the
        // sample, what was laid out behind it is dropped.  Same visits, same arithmetic (visit_geom).
        bool plain = true;
        while (has_ray && t < far && step < max_n_step && visits < FRAME_LANE_VISITS) {
            if (plain || !spec) {
                const Probe p = probe_at(r, cfg, grid, t);
                if (p.occ) {
                    out_t[step] = t;
                    if (EDIT) out_e[step] = (edit_grid[p.index >> 3] >> (p.index & 7u)) & 1u;
                    t += p.dt;
                    step++;
                } else t = skip_to(cfg, t, p.tt);
                visits++;
                plain = p.occ;                             // inside the surface: keep walking visit by visit
                continue;
            }
            float ts[FRAME_SPEC], dts[FRAME_SPEC];
            uint32_t idx[FRAME_SPEC];
            uint8_t byte[FRAME_SPEC];
            float tn = t;
            int nv = 0;
#pragma unroll
            for (int j = 0; j < FRAME_SPEC; j++) {
                if (tn < far) {
                    const VisitGeom v = visit_geom(r, cfg, tn);
                    ts[j] = tn; dts[j] = v.dt; idx[j] = v.index;
                    byte[j] = grid[v.index >> 3];
                    tn = skip_to(cfg, tn, v.tt_empty);
                    nv = j + 1;
                } else { ts[j] = tn; dts[j] = 0.f; idx[j] = 0; byte[j] = 0; }
            }
            bool hit = false;
#pragma unroll
            for (int j = 0; j < FRAME_SPEC; j++) {
                if (!hit && j < nv && ((byte[j] >> (idx[j] & 7u)) & 1u)) {
                    out_t[step] = ts[j];
                    if (EDIT) out_e[step] = (edit_grid[idx[j] >> 3] >> (idx[j] & 7u)) & 1u;
                    t = ts[j] + dts[j];
                    step++;
                    hit = true;
                    plain = true;
                }
            }
            if (!hit) t = tn;
            visits += (uint32_t)nv;
        }
        bool unfinished = has_ray && t < far && step < max_n_step;
        // nothing ahead (see k_frame_coarse_mark): the walk would reach `far` without another sample.  Asked only after a
        // round that found no sample: a lane inside a surface or about to enter one would ask in vain.
        if (cdist && unfinished && step == step_before && frame_clear_octant(r, cfg.bound, t, cdist)) { t = far; unfinished = false; }
        unsigned long long um = __ballot(unfinished);
        if (um == 0ull) break;
        // most of the wave is in transit: lanes are well used.  (A wave whose 64 neighbouring rays all leave a surface
        // together walks on for 16-25 rounds, 130-200 us, while the median wave is done after 11 us -- but handing such
        // waves to the finishing kernel after max_rounds rounds costs more than it saves: one wave per ray there is ~7x
        // the work of a lane here.  Round 4, handing over after 1 / 2 / 3 lane rounds: 15.3 / 15.1 / 14.5 ms per 800x800 frame
        // against 11.9, 92 against 71 ms at 1080p.)
        if (__builtin_popcountll(um) > coop_max && !(q_tasks && st_rounds >= (uint32_t)max_rounds)) {
            // A wave whose rays left a surface together keeps every lane busy, but it is as slow as one ray's walk: ~1 us per visit,
            // 100+ us to the next surface -- and in a frame's late iterations (few rays alive, most of the chip idle) the whole
            // iteration waits for it (a shard of the 1080p frame: iterations 15-47 waited 65 us each on average for lookaheads of
            // 100-140 us while encoder + head took 150 -> 30 us).  One wave per ray walks ~14 cells per microsecond at 7x the
            // work, so such a wave hands ALL its unfinished rays over after `admit_round` rounds -- as long as the launch's total
            // of rays admitted this way stays below `admit_cap` (early iterations have thousands of such waves: they walk on).
            if (!(q_tasks && admit_cap && !admit_tried && st_rounds >= (uint32_t)admit_round)) continue;
            admit_tried = true;
            uint32_t before = 0;
            if (lane == 0) before = atomicAdd(q_count + 2, (uint32_t)__builtin_popcountll(um));
            before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
            if (before + (uint32_t)__builtin_popcountll(um) > admit_cap) continue;
        }
        LOOK_NOTE(2, wall_clock64());
        if (q_tasks) {
            // few stragglers: hand them to k_frame_lookahead_finish (one wave per ray, every SIMD of the chip) instead of
            // finishing them here one after another: a ray leaving the surface for `far` is ~15 candidate passes of ~1 us,
            // up to 8 such rays in one wave were the kernel's 110-150 us while 80 % of its waves had left after 10
            uint32_t base = 0;
            const uint32_t cnt = (uint32_t)__builtin_popcountll(um);
            if (lane == 0) base = atomicAdd(q_count, cnt);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (unfinished) {
                const uint32_t slot = base + (uint32_t)__builtin_popcountll(um & ((1ull << lane) - 1ull));
                q_tasks[slot] = LookTask{index, step, t};
                has_ray = false;                           // its count / end time are the finishing kernel's to write
            }
            break;
        }
        while (um) {                                       // few stragglers: finish each with the whole wave
            st_coop++;
            const int L = __builtin_ctzll(um);
            um &= um - 1ull;
            Ray rl;
            rl.ox = bcast(r.ox, L); rl.oy = bcast(r.oy, L); rl.oz = bcast(r.oz, L);
            rl.dx = bcast(r.dx, L); rl.dy = bcast(r.dy, L); rl.dz = bcast(r.dz, L);
            rl.rdx = bcast(r.rdx, L); rl.rdy = bcast(r.rdy, L); rl.rdz = bcast(r.rdz, L);
            const uint32_t step_l = (uint32_t)__builtin_amdgcn_readlane((int)step, L);
            const uint32_t index_l = (uint32_t)__builtin_amdgcn_readlane((int)index, L);
            float t_end;
            const uint32_t got = frame_lookahead_coop<EDIT>(rl, cfg, grid, edit_grid, bcast(t, L), bcast(far, L), max_n_step - step_l,
                                                            out.t + (size_t)index_l * FRAME_LA + step_l,
                                                            EDIT ? out.e + (size_t)index_l * FRAME_LA + step_l : nullptr, lane, t_end, cdist);
            if (lane == L) { step += got; t = t_end; }
        }
        break;
    }
    if (has_ray) { out.cnt[index] = step; out.tend[index] = t; }
    LOOK_NOTE(3, wall_clock64());
    LOOK_NOTE(4, (unsigned long long)st_rounds | ((unsigned long long)st_coop << 16) | ((unsigned long long)__builtin_popcountll(__ballot(has_ray)) << 32));
}

// the stragglers of one k_frame_lookahead launch, one wave per ray (frame_lookahead_coop); workgroup 0 also clears the OTHER
// task counter for the next iteration's lane phase
template <bool EDIT>
__global__ __launch_bounds__(FRAME_BLOCK) void k_frame_lookahead_finish(
    uint32_t max_n_step, const uint32_t* __restrict__ q_count, uint32_t* __restrict__ q_count_next, const LookTask* __restrict__ q_tasks,
    LookRec out, const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ fars, MarchCfg cfg,
    const uint8_t* __restrict__ grid, const uint8_t* __restrict__ edit_grid, const uint8_t* __restrict__ cdist) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { q_count_next[0] = 0u; q_count_next[2] = 0u; }   // task count and admitted-rays count of the next launch
    const uint32_t n_tasks = *q_count;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (FRAME_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (uint32_t i = wave; i < n_tasks; i += gridDim.x * (FRAME_BLOCK / 64)) {
        const LookTask tk = q_tasks[i];
        const uint32_t index = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk.index), step = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk.step);
        const float t = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tk.t)));
        const Ray r = load_ray(rays_o, rays_d, index);
        float t_end;
        const uint32_t got = frame_lookahead_coop<EDIT>(r, cfg, grid, edit_grid, t, fars[index], max_n_step - step,
                                                        out.t + (size_t)index * FRAME_LA + step,
                                                        EDIT ? out.e + (size_t)index * FRAME_LA + step : nullptr, lane, t_end, cdist);
        if (lane == 0) { out.cnt[index] = step + got; out.tend[index] = t_end; }
    }
}

// march_rays (raymarching.cu:700-805 / :811-926) as a replay of the recorded sample times; see the section comment.
// The survivors of the previous iteration arrive as segments (frame_locate); this kernel lists them in order (alive[]),
// publishes the loop state and writes the sample rows.  It also tells the lookahead stream when to start (k_frame_wait there
// polls `go`): at its beginning (go_early: the lookahead of this iteration needs nothing this kernel writes -- only that the
// previous head kernel is complete, which this kernel running proves) or when its last-dispatched workgroup is done (the
// lookahead then does not compete with this kernel, which is on the caller's critical path, for the memory system).
// rows of 64 rays with n_step samples each + the padding of the groups they touch (a multiple of 16: the edit flags follow the
// floats, and every wave's image stays 16-byte aligned)
__host__ __device__ constexpr uint32_t emit_img_rows(uint32_t n_step) { return (64u * n_step + 4u * (64u / (64u / n_step) + 2u) + 15u) & ~15u; }
static_assert(emit_img_rows(3) % 16u == 0u && emit_img_rows(8) >= 64u * 8u + 4u * 10u, "image sizing");
// xyz + delta rows (20 B each), 64 x (direction, samples held), edit flags
__host__ __device__ constexpr uint32_t emit_img_floats(uint32_t rows) { return 5u * rows + 256u + rows / 4u; }
template <bool EDIT>
__global__ __launch_bounds__(FRAME_BLOCK) void k_frame_emit(
    const FrameCtrl* __restrict__ prev, FrameCtrl* __restrict__ cur, FrameSegs sg, uint32_t N, uint32_t row_budget, uint32_t max_steps,
    uint32_t max_n_step, int32_t* __restrict__ alive, LookRec in, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, MarchCfg cfg, float* __restrict__ xyzs, float* __restrict__ dirs,
    float* __restrict__ deltas, uint8_t* __restrict__ edit_occ, const float* __restrict__ noises,
    FrameMirror* __restrict__ mirror, uint64_t frame_id, unsigned long long* __restrict__ go, unsigned long long go_value, int go_early,
    int use_lds, uint32_t img_rows) {
    const int lane = threadIdx.x & 63;
    const uint32_t wv = blockIdx.x * (FRAME_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (go && go_early && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(go, go_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const FrameSlot f = frame_locate(sg, N, wv, lane);
    const FrameCtrl c = frame_next_ctrl(*prev, f.n_alive, row_budget, max_steps, max_n_step);
    const uint32_t rpg = lae::frame_rays_per_group(max(c.n_step, 1u));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *cur = c;
        mirror->total_rows = c.total_rows; mirror->iters = c.iter; mirror->n_alive = c.n_alive; mirror->done = c.done;   // done last
        __threadfence_system();
        mirror->tag = frame_id;
        for (uint32_t row = c.n_rows; row < ((c.n_rows + 15u) & ~15u); row++) {     // pad rows of the last 16-row MLP tile
            xyzs[3 * (size_t)row] = 0.f; xyzs[3 * (size_t)row + 1] = 0.f; xyzs[3 * (size_t)row + 2] = 0.f;
            dirs[3 * (size_t)row] = 0.f; dirs[3 * (size_t)row + 1] = 0.f; dirs[3 * (size_t)row + 2] = 0.f;
            deltas[2 * (size_t)row] = 0.f; deltas[2 * (size_t)row + 1] = 0.f;
        }
    }
    // n_step >= 3: a lane's rows are n_step x 32 bytes apart from its neighbour's, every store instruction of the loop below
    // scatters 64 x 12 (or 8) bytes over 64 x 32 x n_step bytes (the emit kernel of the late iterations, 8 rows per ray, took 28 us
    // for the rows the early ones write in 11).  The wave's rays are consecutive in the compact order, so their rows (and the
    // padding rows between them) are ONE contiguous range: the lanes build it in a wave-private LDS image and the wave
    // stores it with consecutive lanes on consecutive words.
    extern __shared__ float emit_lds[];
    // (img_rows: capacity of a wave's image as the host sized it from its lagging bound of n_alive; an iteration whose n_step
    // outgrew it stores straight from the lanes)
    if (use_lds && c.n_step >= (uint32_t)use_lds && !c.done && 64u * c.n_step + 4u * (64u / rpg + 2u) <= img_rows) {
        const unsigned long long hm = __ballot(f.has);
        if (hm) {                                              // valid lanes are a prefix of the wave (frame_locate)
            const uint32_t cnt = (uint32_t)__builtin_popcountll(hm), n_step = c.n_step;
            float* img = emit_lds + (size_t)(threadIdx.x >> 6) * emit_img_floats(img_rows);
            float* ix = img; float* il = img + 3 * img_rows; float* idr = img + 5 * img_rows;   // xyz rows, delta rows, per RAY: direction + samples held
            uint8_t* ie = reinterpret_cast<uint8_t*>(img + 5 * img_rows + 256);
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)f.n);
            const uint32_t g0 = n0 / rpg, row0 = g0 * 64u + (n0 - g0 * rpg) * n_step;
            const uint32_t nl = n0 + cnt - 1u, gl = nl / rpg, sl = nl - gl * rpg;
            const uint32_t row1 = sl == rpg - 1u ? gl * 64u + 64u : gl * 64u + (sl + 1u) * n_step;   // the last slot of a group takes its padding rows along
            const uint32_t nrows = row1 - row0;                // <= 64 n_step + 4 (groups + 1) <= img_rows
            if (f.has) {
                const uint32_t index = f.index;
                alive[f.n] = (int32_t)index;
                const Ray r = load_ray(rays_o, rays_d, index);
                const uint32_t have = min(in.cnt[index], n_step);
                float last_t = in.tc[index];
                if (sg.nu == 0) last_t = perturbed_start(cfg, last_t, noises, f.n);
                const float4 ra = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[0];
                const float4 rb = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[1];
                const float st[FRAME_LA] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                unsigned long long ste = 0ull;
                if (EDIT) ste = *reinterpret_cast<const unsigned long long*>(in.e + (size_t)index * FRAME_LA);
                const uint32_t grp = f.n / rpg, slot = f.n - grp * rpg;
                uint32_t lr = grp * 64u + slot * n_step - row0;
                idr[4 * lane] = r.dx; idr[4 * lane + 1] = r.dy; idr[4 * lane + 2] = r.dz; idr[4 * lane + 3] = __uint_as_float(have);
#pragma unroll
                for (uint32_t j = 0; j < FRAME_LA; j++, lr++) {
                    if (j >= n_step) break;                    // uniform
                    const bool real = j < have;                // the reference's buffers are torch.zeros
                    const float t = st[j], dt = step_of(cfg, t), tn = t + dt;
                    ix[3 * lr] = real ? clampf(fmaf(t, r.dx, r.ox), -cfg.bound, cfg.bound) : 0.f;
                    ix[3 * lr + 1] = real ? clampf(fmaf(t, r.dy, r.oy), -cfg.bound, cfg.bound) : 0.f;
                    ix[3 * lr + 2] = real ? clampf(fmaf(t, r.dz, r.oz), -cfg.bound, cfg.bound) : 0.f;
                    il[2 * lr] = real ? dt : 0.f; il[2 * lr + 1] = real ? tn - last_t : 0.f;
                    if (real) last_t = tn;
                    if (EDIT) ie[lr] = real ? (uint8_t)(ste >> (8u * j)) : (uint8_t)0;
                }
                if (slot == rpg - 1u)                          // the group's padding rows (n_step = 3, 5, 6, 7: at most 4)
                    for (uint32_t pr = rpg * n_step; pr < 64u; pr++) {
                        const uint32_t q = grp * 64u + pr - row0;
                        ix[3 * q] = 0.f; ix[3 * q + 1] = 0.f; ix[3 * q + 2] = 0.f; il[2 * q] = 0.f; il[2 * q + 1] = 0.f;
                        if (EDIT) ie[q] = 0;
                    }
            }
            __builtin_amdgcn_wave_barrier();                    // wave-private image: one wave's LDS accesses execute in order, nothing to wait for
            float* ox = xyzs + 3 * (size_t)row0; float* od = dirs + 3 * (size_t)row0; float* ol = deltas + 2 * (size_t)row0;
            const uint32_t inv_ns = 65536u / n_step + 1u;      // pos / n_step for pos < 64 (checked exhaustively)
            for (uint32_t e = (uint32_t)lane; e < nrows * 3u; e += 64u) {
                ox[e] = ix[e];
                const uint32_t lr = (e * 21846u) >> 16, comp = e - 3u * lr;        // e / 3 for e < 4096
                const uint32_t row = row0 + lr, pos = row & 63u, slot = (pos * inv_ns) >> 16, j = pos - slot * n_step;
                const uint32_t rl = min((row >> 6) * rpg + min(slot, rpg - 1u) - n0, 63u);
                const float4 dh = reinterpret_cast<const float4*>(idr)[rl];
                const float dv = comp == 0u ? dh.x : comp == 1u ? dh.y : dh.z;
                od[e] = (slot < rpg && j < __float_as_uint(dh.w)) ? dv : 0.f;
            }
            for (uint32_t e = (uint32_t)lane; e < nrows * 2u; e += 64u) ol[e] = il[e];
            if (EDIT) for (uint32_t e = (uint32_t)lane; e < nrows; e += 64u) edit_occ[row0 + e] = ie[e];
        }
    } else
    if (!c.done && f.has) {
    const uint32_t n = f.n, n_step = c.n_step;
    const uint32_t index = f.index;
    alive[n] = (int32_t)index;
    const Ray r = load_ray(rays_o, rays_d, index);
    const uint32_t have = min(in.cnt[index], n_step);
    float last_t = in.tc[index];
    if (sg.nu == 0) last_t = perturbed_start(cfg, last_t, noises, n);
    // the record as two 16-byte loads (+ one 8-byte load of the edit flags), not a dependent load per sample
    const float4 ra = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[0];
    const float4 rb = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[1];
    const float st[FRAME_LA] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    unsigned long long ste = 0ull;
    if (EDIT) ste = *reinterpret_cast<const unsigned long long*>(in.e + (size_t)index * FRAME_LA);
    const uint32_t grp = n / rpg, slot = n - grp * rpg;
    size_t row = (size_t)grp * 64u + (size_t)slot * n_step;
#pragma unroll
    for (uint32_t j = 0; j < FRAME_LA; j++, row++) {
        if (j >= n_step) break;                            // uniform
        if (j < have) {                                    // :761-790 with the recorded time
            const float t = st[j], dt = step_of(cfg, t), tn = t + dt;
            xyzs[3 * row] = clampf(fmaf(t, r.dx, r.ox), -cfg.bound, cfg.bound);
            xyzs[3 * row + 1] = clampf(fmaf(t, r.dy, r.oy), -cfg.bound, cfg.bound);
            xyzs[3 * row + 2] = clampf(fmaf(t, r.dz, r.oz), -cfg.bound, cfg.bound);
            dirs[3 * row] = r.dx; dirs[3 * row + 1] = r.dy; dirs[3 * row + 2] = r.dz;
            deltas[2 * row] = dt; deltas[2 * row + 1] = tn - last_t; last_t = tn;
            if (EDIT) edit_occ[row] = (uint8_t)(ste >> (8u * j));
        } else {                                           // the reference's buffers are torch.zeros
            xyzs[3 * row] = 0.f; xyzs[3 * row + 1] = 0.f; xyzs[3 * row + 2] = 0.f;
            dirs[3 * row] = 0.f; dirs[3 * row + 1] = 0.f; dirs[3 * row + 2] = 0.f;
            deltas[2 * row] = 0.f; deltas[2 * row + 1] = 0.f;
            if (EDIT) edit_occ[row] = 0;
        }
    }
    if (slot == rpg - 1u) {                                // the group's padding rows (n_step = 3, 5, 6, 7: at most 4)
        for (size_t pr = (size_t)grp * 64u + (size_t)rpg * n_step; pr < (size_t)grp * 64u + 64u; pr++) {
            xyzs[3 * pr] = 0.f; xyzs[3 * pr + 1] = 0.f; xyzs[3 * pr + 2] = 0.f;
            dirs[3 * pr] = 0.f; dirs[3 * pr + 1] = 0.f; dirs[3 * pr + 2] = 0.f;
            deltas[2 * pr] = 0.f; deltas[2 * pr + 1] = 0.f;
            if (EDIT) edit_occ[pr] = 0;
        }
    }
    }
    if (go && !go_early) {
        // the workgroup holding the LAST ray of the list was dispatched last of those with work: when it is done the kernel
        // is (all but) done and the lookahead may start.  Nothing alive / loop over: workgroup 0 says so.
        const int last = __syncthreads_or((f.has && f.n + 1u == f.n_alive) || ((c.done || f.n_alive == 0u) && blockIdx.x == 0));
        if (last && threadIdx.x == 0) { __threadfence(); __hip_atomic_store(go, go_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

// k_frame_emit for iterations whose n_step is CERTAINLY 8 (the host's bound of n_alive already gives budget / bound >= 8, and 8
// is the cap): rows of ray n are 8 n .. 8 n + 7, no padding.  The general kernel's LDS image for 64 rays x 8 samples is 13 KB per wave --
// three workgroups per CU, 768 resident of the 1024 such a launch has: it ran in two rounds (22-24 us against 12-15 for n_step
// 3-7).  Here a wave's rays go through a 32-ray image in two passes (6 KB per wave).  A separate kernel on purpose: the same two
// passes inside k_frame_emit changed its register allocation and slowed EVERY path.
constexpr uint32_t EMIT8_ROWS = 256;                       // 32 rays x 8 samples
constexpr uint32_t EMIT8_FLOATS = 5u * EMIT8_ROWS + 128u + EMIT8_ROWS / 4u;   // xyz + delta rows, 32 x (direction, samples held), edit flags
template <bool EDIT>
__global__ __launch_bounds__(FRAME_BLOCK) void k_frame_emit8(
    const FrameCtrl* __restrict__ prev, FrameCtrl* __restrict__ cur, FrameSegs sg, uint32_t N, uint32_t row_budget, uint32_t max_steps,
    uint32_t max_n_step, int32_t* __restrict__ alive, LookRec in, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, MarchCfg cfg, float* __restrict__ xyzs, float* __restrict__ dirs,
    float* __restrict__ deltas, uint8_t* __restrict__ edit_occ, const float* __restrict__ noises,
    FrameMirror* __restrict__ mirror, uint64_t frame_id, unsigned long long* __restrict__ go, unsigned long long go_value, int go_early) {
    __shared__ float emit8_lds[(FRAME_BLOCK / 64) * EMIT8_FLOATS];
    const int lane = threadIdx.x & 63;
    const uint32_t wv = blockIdx.x * (FRAME_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (go && go_early && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(go, go_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const FrameSlot f = frame_locate(sg, N, wv, lane);
    const FrameCtrl c = frame_next_ctrl(*prev, f.n_alive, row_budget, max_steps, max_n_step);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *cur = c;
        mirror->total_rows = c.total_rows; mirror->iters = c.iter; mirror->n_alive = c.n_alive; mirror->done = c.done;   // done last
        __threadfence_system();
        mirror->tag = frame_id;
        for (uint32_t row = c.n_rows; row < ((c.n_rows + 15u) & ~15u); row++) {     // pad rows of the last 16-row MLP tile
            xyzs[3 * (size_t)row] = 0.f; xyzs[3 * (size_t)row + 1] = 0.f; xyzs[3 * (size_t)row + 2] = 0.f;
            dirs[3 * (size_t)row] = 0.f; dirs[3 * (size_t)row + 1] = 0.f; dirs[3 * (size_t)row + 2] = 0.f;
            deltas[2 * (size_t)row] = 0.f; deltas[2 * (size_t)row + 1] = 0.f;
        }
    }
    const unsigned long long hm = __ballot(f.has);
    if (!c.done && c.n_step == 8u && hm) {                     // (n_step == 8 by construction of the launch; valid lanes are a prefix of the wave)
        const uint32_t cnt = (uint32_t)__builtin_popcountll(hm);
        float* img = emit8_lds + (size_t)(threadIdx.x >> 6) * EMIT8_FLOATS;
        float* ix = img; float* il = img + 3 * EMIT8_ROWS; float* idr = img + 5 * EMIT8_ROWS;
        uint8_t* ie = reinterpret_cast<uint8_t*>(img + 5 * EMIT8_ROWS + 128);
        const uint32_t n_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)f.n);
        Ray r{};
        uint32_t have = 0;
        float last_t = 0.f;
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        unsigned long long ste = 0ull;
        if (f.has) {
            const uint32_t index = f.index;
            alive[f.n] = (int32_t)index;
            r = load_ray(rays_o, rays_d, index);
            have = min(in.cnt[index], 8u);
            last_t = in.tc[index];
            if (sg.nu == 0) last_t = perturbed_start(cfg, last_t, noises, f.n);
            ra = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[0];
            rb = reinterpret_cast<const float4*>(in.t + (size_t)index * FRAME_LA)[1];
            if (EDIT) ste = *reinterpret_cast<const unsigned long long*>(in.e + (size_t)index * FRAME_LA);
        }
        const float st[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll 1
        for (uint32_t h = 0; h < 2u; h++) {
            const uint32_t lane_lo = h * 32u;
            if (lane_lo >= cnt) break;
            const uint32_t cnt_h = min(cnt - lane_lo, 32u), row0 = (n_first + lane_lo) * 8u, nrows = cnt_h * 8u;
            if (f.has && (uint32_t)lane >= lane_lo && (uint32_t)lane < lane_lo + 32u) {
                const uint32_t li = (uint32_t)lane - lane_lo;
                idr[4 * li] = r.dx; idr[4 * li + 1] = r.dy; idr[4 * li + 2] = r.dz; idr[4 * li + 3] = __uint_as_float(have);
                float lt = last_t;
#pragma unroll
                for (uint32_t j = 0; j < 8u; j++) {
                    const uint32_t lr = li * 8u + j;
                    const bool real = j < have;                // the reference's buffers are torch.zeros
                    const float t = st[j], dt = step_of(cfg, t), tn = t + dt;
                    ix[3 * lr] = real ? clampf(fmaf(t, r.dx, r.ox), -cfg.bound, cfg.bound) : 0.f;
                    ix[3 * lr + 1] = real ? clampf(fmaf(t, r.dy, r.oy), -cfg.bound, cfg.bound) : 0.f;
                    ix[3 * lr + 2] = real ? clampf(fmaf(t, r.dz, r.oz), -cfg.bound, cfg.bound) : 0.f;
                    il[2 * lr] = real ? dt : 0.f; il[2 * lr + 1] = real ? tn - lt : 0.f;
                    if (real) lt = tn;
                    if (EDIT) ie[lr] = real ? (uint8_t)(ste >> (8u * j)) : (uint8_t)0;
                }
            }
            __builtin_amdgcn_wave_barrier();                    // wave-private image: one wave's LDS accesses execute in order
            float* ox = xyzs + 3 * (size_t)row0; float* od = dirs + 3 * (size_t)row0; float* ol = deltas + 2 * (size_t)row0;
            for (uint32_t e = (uint32_t)lane; e < nrows * 3u; e += 64u) {
                ox[e] = ix[e];
                const uint32_t lr = (e * 21846u) >> 16, comp = e - 3u * lr;        // e / 3 for e < 4096
                const float4 dh = reinterpret_cast<const float4*>(idr)[lr >> 3];
                const float dv = comp == 0u ? dh.x : comp == 1u ? dh.y : dh.z;
                od[e] = (lr & 7u) < __float_as_uint(dh.w) ? dv : 0.f;
            }
            for (uint32_t e = (uint32_t)lane; e < nrows * 2u; e += 64u) ol[e] = il[e];
            if (EDIT) for (uint32_t e = (uint32_t)lane; e < nrows; e += 64u) edit_occ[row0 + e] = ie[e];
            __builtin_amdgcn_wave_barrier();                    // the second pass overwrites the image
        }
    }
    if (go && !go_early) {
        const int last = __syncthreads_or((f.has && f.n + 1u == f.n_alive) || ((c.done || f.n_alive == 0u) && blockIdx.x == 0));
        if (last && threadIdx.x == 0) { __threadfence(); __hip_atomic_store(go, go_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

// renderer.py:381-383: the accumulators go out to the caller's arrays with the background blend and the depth normalisation
__global__ void k_frame_finish(uint32_t N, const float* __restrict__ nears, const float* __restrict__ fars,
                               const RayAcc* __restrict__ acc, float* __restrict__ weights_sum, float* __restrict__ depth,
                               float* __restrict__ image, float* __restrict__ weights_edit, float* __restrict__ depth_edit,
                               const float* __restrict__ bg_rays, float bg_r, float bg_g, float bg_b, int blend_bg, int scale_depth,
                               const uint32_t* __restrict__ hung_dev, uint32_t hung_tag) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (*hung_dev == hung_tag) {                           // a cross-stream wait of this frame gave up: never hand out what was composited
        const float q = __builtin_nanf("");
        weights_sum[n] = q; depth[n] = q;
        image[3 * (size_t)n] = q; image[3 * (size_t)n + 1] = q; image[3 * (size_t)n + 2] = q;
        if (weights_edit) { weights_edit[n] = q; depth_edit[n] = q; }
        return;
    }
    const float4 a0 = reinterpret_cast<const float4*>(acc + n)[0], a1 = reinterpret_cast<const float4*>(acc + n)[1];
    float r = a0.z, g = a0.w, b = a1.x, d = a0.y;
    if (blend_bg) {
        const float om = 1 - a0.x;
        const float b0 = bg_rays ? bg_rays[3 * (size_t)n] : bg_r, b1 = bg_rays ? bg_rays[3 * (size_t)n + 1] : bg_g,
                    b2 = bg_rays ? bg_rays[3 * (size_t)n + 2] : bg_b;
        r += om * b0; g += om * b1; b += om * b2;
    }
    if (scale_depth) d = fmaxf(d - nears[n], 0.0f) / (fars[n] - nears[n]);
    weights_sum[n] = a0.x; depth[n] = d;
    image[3 * (size_t)n] = r; image[3 * (size_t)n + 1] = g; image[3 * (size_t)n + 2] = b;
    if (weights_edit) { weights_edit[n] = a1.z; depth_edit[n] = a1.w; }
}

}  // namespace

extern "C" {

// ---- whole-frame inference (MI355X-native; replaces the host loop of NeRFRenderer.run_cuda, renderer.py:335-387,
// and of run_cuda_distill, :394-480, when edit_grid != NULL)
static inline uint64_t frame_budget(uint32_t N, uint64_t row_budget) { return row_budget < N ? N : std::min<uint64_t>(row_budget, 0xe0000000ull); }
// rows of an iteration in the grouped layout (lae_common.h FrameCtrl): at most 64 / 60 of the budget (n_step = 7: 63 of
// 64 rows of a group hold samples; n_step = 5, 6: 60) plus one partial group
static inline uint64_t frame_padded_rows(uint64_t rows) { return (rows + rows / 15 + 64 + 63) / 64 * 64; }
static inline uint64_t frame_cap(uint64_t budget) { return frame_padded_rows(budget) + 64; }
static inline uint64_t al256(uint64_t b) { return (b + 255) / 256 * 256; }
static inline uint64_t frame_seg_elems(uint32_t N) { return (uint64_t)N + (uint64_t)FRAME_SEG_MAX * FRAME_SEG_SLACK; }
uint64_t lae_render_frame_workspace_bytes(uint32_t N, uint32_t L, uint64_t row_budget) {
    const uint64_t cap = frame_cap(frame_budget(N, row_budget));
    return 256 /*ctrl x2*/ + 2 * al256(4ull * FRAME_SEG_MAX) /*segment counts x2*/ + 2 * al256(4ull * FRAME_SEG_MAX) /*workgroup sums x2*/ +
           al256(4ull * N) /*alive*/ + 2 * al256(4 * frame_seg_elems(N)) /*survivor segments x2*/ + al256(32ull * N) /*accumulators*/ +
           2 * al256(4ull * N) /*nears, fars*/ +
           2 * (al256(4ull * FRAME_LA * N) + al256((uint64_t)FRAME_LA * N) + 3 * al256(4ull * N)) /*lookahead records x2: times, edit flags, count, end t, tc*/ +
           2 * al256(12 * cap) /*xyzs, dirs*/ + al256(8 * cap) /*deltas*/ + al256(cap) /*edit_occ*/ +
           al256((uint64_t)L * cap * 4) /*features [L,cap,2] fp16*/ +
           256 /*straggler task counters x2*/ + al256(12ull * N) /*straggler tasks*/ +
           al256(4ull * FRAME_CG * FRAME_CG) + al256(2ull * FRAME_CG * FRAME_CG * FRAME_CG) /*coarse mask, distance field + octant flags*/;
}

namespace {
struct FrameHost {                                        // process-wide helpers of the frame loop, created on first use
    FrameMirror* mirror_h = nullptr;
    FrameMirror* mirror_d = nullptr;
    hipStream_t side = nullptr;                           // lookahead marcher runs here, beside the encoder / MLP kernels
    unsigned long long* flags = nullptr;                  // device: [0] "lookahead may start" (emit kernel), [1] "lookahead done" (side stream); values only grow
    uint64_t frame_counter = 0;
    hipStream_t last_stream = nullptr;                    // stream of the previous frame (its tail may still be queued)
    bool have_last = false;
    bool ok = false;
    // flags[2] (as uint32): tag of the frame a wait gave up in; flags[3]: the probe's word
    int last_mode = -1;                                   // how the most recent frame ran: 1 overlapped, 0 in line, -1 no frame yet
    bool degraded = false;                                // a frame's handshake gave up once: every later frame runs its lookahead in line
    bool warned = false;
    hipStream_t probed[8] = {};                           // caller streams that were probed, the verdicts and the side stream chosen for each
    bool probed_ok[8] = {};
    hipStream_t probed_side[8] = {};
    uint32_t probed_age[8] = {};                          // frames of that caller stream since its verdict was made
    uint8_t probed_retries[8] = {};                       // re-probes of a failed verdict so far
    int n_probed = 0;
    uint64_t probe_counter = 0;
    static constexpr int MAX_CAND = 5;
    hipStream_t cand[MAX_CAND] = {};                      // side-stream candidates: [0] the configured priority (default highest), [1..] the caller's class, made on demand
    int n_cand = 0;
    int later_prio = 0;                                   // priority of candidates 1..
    float last_probe_us[MAX_CAND] = {-1.f, -1.f, -1.f, -1.f, -1.f};   // handshake time of the most recent probe per candidate (lae_render_frame_probe_us)
    bool init() {
        if (ok) return true;
        void* hp = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&hp, 256, hipHostMallocMapped) != hipSuccess) return false;
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipHostFree(hp); return false; }
        memset(hp, 0, 256);
        mirror_h = reinterpret_cast<FrameMirror*>(hp); mirror_d = reinterpret_cast<FrameMirror*>(dp);
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
        int prio = hi;
        const char* e = getenv("LAE_FRAME_SIDE_PRIO");    // 1 lowest / 0 normal / -1 highest: ONE candidate of that class (A/B, tests)
        if (e) { const int m = atoi(e); prio = m > 0 ? lo : m == 0 ? (lo + hi) / 2 : hi; }
        if (hipStreamCreateWithPriority(&cand[0], hipStreamNonBlocking, prio) != hipSuccess) return false;
        n_cand = 1;
        later_prio = e ? prio : (lo + hi) / 2;            // further candidates (frame_pick_side makes them when candidate 0 is slow): the caller's class
        side = cand[0];
        // The two streams hand each other work through two 64-bit words in device memory, not through events (k_frame_wait /
        // k_frame_signal above): an event record + wait is 10-12 us per dependency here, 2-3 of them per iteration.
        if (hipMalloc(reinterpret_cast<void**>(&flags), 256) != hipSuccess) return false;
        if (hipMemset(flags, 0, 256) != hipSuccess) return false;
        ok = true;
        return true;
    }
};
FrameHost g_frame;
std::mutex g_frame_mtx;
int g_frame_overlap = 1;                                  // 0: lookahead in-line on the caller's stream (A/B switch)
uint32_t frame_wait_spins() {                             // polls before a k_frame_wait gives up; LAE_FRAME_WAIT_SPINS_LOG2 shortens it for the tests
    static const uint32_t v = [] { const char* e = getenv("LAE_FRAME_WAIT_SPINS_LOG2"); const int l = e ? atoi(e) : 23; return 1u << std::min(std::max(l, 8), 26); }();
    return v;
}
void frame_warn_once(const char* why) {
    if (g_frame.warned) return;
    g_frame.warned = true;
    fprintf(stderr, "laenerf_amd: render_frame: %s; the lookahead marcher runs in line on the caller's stream from now on "
                    "(same image bit for bit, ~10-25 %% slower frames)\n", why);
}
// Can the caller's stream and the side stream make progress side by side?  The handshake below needs it: a wait kernel on one
// stream polls a word a kernel on the other stream stores.  Not when both map to one hardware queue (GPU_MAX_HW_QUEUES=1, many
// live streams of one priority), when dispatch is serialised (AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING) or under a profiler
// that collects counters.  Probed once per caller stream: a wait on `s`, THEN a signal on the side stream; ~15 us when it
// passes, ~40 ms once when it does not.
static bool frame_probe_concurrent(hipStream_t s, hipStream_t side, uint32_t spins = 1u << 15) {
    volatile uint32_t* res_h = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<uint8_t*>(g_frame.mirror_h) + 128);
    volatile uint32_t* res_d = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<uint8_t*>(g_frame.mirror_d) + 128);
    *res_h = 0u;
    const unsigned long long v = ++g_frame.probe_counter;
    k_frame_probe_wait<<<1, 64, 0, s>>>(g_frame.flags + 3, v, res_d, spins);
    k_frame_signal<<<1, 1, 0, side>>>(g_frame.flags + 3, v);
    return hipStreamSynchronize(side) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && *res_h == 1u;
}
// Microseconds for FRAME_PROBE_TRIPS hand-overs caller -> side -> caller through the loop's own mechanism (a one-thread store on
// one stream, a polling wait on the other; everything queued up front, timed on the device between two events on `s`); < 0 on
// failure.  Only called for streams that passed frame_probe_concurrent (a wait would otherwise run into its bound).
constexpr int FRAME_PROBE_TRIPS = 16;
static float frame_probe_handshake_us(hipStream_t s, hipStream_t side) {
    volatile uint32_t* res_d = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<uint8_t*>(g_frame.mirror_d) + 132);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); return -1.f; }
    unsigned long long* fa = g_frame.flags + 3;          // caller waits, side stores
    unsigned long long* fb = g_frame.flags + 4;          // side waits, caller stores (values only grow, like flags[3])
    const unsigned long long base = g_frame.probe_counter;
    g_frame.probe_counter += FRAME_PROBE_TRIPS;
    bool ok = hipEventRecord(e0, s) == hipSuccess;
    for (int i = 1; i <= FRAME_PROBE_TRIPS && ok; i++) {
        if (i > 1) k_frame_probe_wait<<<1, 64, 0, side>>>(fb, base + i - 1, res_d, 1u << 15);
        k_frame_signal<<<1, 1, 0, side>>>(fa, base + i);
        k_frame_probe_wait<<<1, 64, 0, s>>>(fa, base + i, res_d, 1u << 15);
        k_frame_signal<<<1, 1, 0, s>>>(fb, base + i);
    }
    ok = ok && hipEventRecord(e1, s) == hipSuccess && hipStreamSynchronize(side) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float ms = -1.f;
    if (ok && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = -1.f;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms < 0.f ? -1.f : ms * 1000.f;
}
// Can the caller's stream and a side stream make progress side by side, and does the side stream pick its work up promptly?  The
// handshake below needs the first: a wait kernel on one stream polls a word a kernel on the other stream stores.  Not when both
// map to one hardware queue (GPU_MAX_HW_QUEUES=1, many live streams of one priority), when dispatch is serialised
// (AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING) or under a profiler that collects counters.  The second decides the frame time:
// the lookahead stream wakes up once per iteration, and a hardware queue that is slow to pick a dispatch up costs ~50 us per
// hand-over instead of ~11 (round 5, MI355X / ROCm 7.2: once three or more other streams had been used in the process the
// highest-priority stream was such a queue -- a 9.8 ms frame took 23 ms -- while a stream of the caller's class was not; with two
// used streams it was the other way round; in a fresh process both are fast).  So, once per caller stream: candidate 0 (the
// configured priority, default highest) is probed -- ~15 us for the concurrency test, two timed runs of 16 hand-overs, ~0.2 ms each --
// and taken if it is concurrent and fast (< FRAME_FAST_US for the 16); otherwise up to four streams of the caller's class are
// created and probed in turn (a new stream lands on another hardware queue) and the first fast one is taken, else the fastest seen.
// ~40 ms once per candidate that does not run beside the caller's stream.
// -> the side stream to use, nullptr: run in line
constexpr float FRAME_FAST_US = 400.f;                    // 16 hand-overs: ~170-200 us on a prompt queue, ~850 on a slow one
// A cached verdict is not for ever: "not concurrent" may be a signal kernel that a busy neighbour process delayed past
// the probe's ~40 ms bound, and a destroyed stream's handle can come back for a new stream.  A failed verdict is probed again
// after FRAME_REPROBE_FAILED frames of that stream with an 8x longer bound, at most three times; a good one after
// FRAME_REPROBE_GOOD frames (~0.5 ms once in a few thousand frames).
constexpr uint32_t FRAME_REPROBE_FAILED = 64, FRAME_REPROBE_GOOD = 4096;
hipStream_t frame_pick_side(hipStream_t s) {
    int reuse = -1;
    uint32_t spins = 1u << 15;
    for (int i = 0; i < g_frame.n_probed; i++)
        if (g_frame.probed[i] == s) {
            const uint32_t age = ++g_frame.probed_age[i];
            if (g_frame.probed_ok[i] && age < FRAME_REPROBE_GOOD) return g_frame.probed_side[i];
            if (!g_frame.probed_ok[i] && (age < FRAME_REPROBE_FAILED || g_frame.probed_retries[i] >= 3)) return nullptr;
            if (!g_frame.probed_ok[i]) { g_frame.probed_retries[i]++; spins = 1u << 18; }
            reuse = i;
            break;
        }
    hipStream_t best = nullptr;
    float best_us = 0.f;
    for (int c = 0; c < FrameHost::MAX_CAND; c++) {
        if (c >= g_frame.n_cand) {
            if (hipStreamCreateWithPriority(&g_frame.cand[c], hipStreamNonBlocking, g_frame.later_prio) != hipSuccess) break;
            g_frame.n_cand = c + 1;
        }
        g_frame.last_probe_us[c] = -1.f;
        if (!frame_probe_concurrent(s, g_frame.cand[c], spins)) continue;
        (void)frame_probe_handshake_us(s, g_frame.cand[c]);                  // first run: the queue wakes up, code objects load
        const float us = frame_probe_handshake_us(s, g_frame.cand[c]);
        g_frame.last_probe_us[c] = us;
        if (us < 0.f) continue;
        if (!best || us < best_us) { best = g_frame.cand[c]; best_us = us; }
        if (us < FRAME_FAST_US) break;
    }
    const int slot = reuse >= 0 ? reuse : g_frame.n_probed < 8 ? g_frame.n_probed++ : (int)(g_frame.probe_counter & 7u);
    if (reuse < 0) g_frame.probed_retries[slot] = 0;
    g_frame.probed[slot] = s; g_frame.probed_ok[slot] = best != nullptr; g_frame.probed_side[slot] = best; g_frame.probed_age[slot] = 0;
    return best;
}
}  // namespace

int lae_render_frame_set_overlap(int on) { g_frame_overlap = on ? 1 : 0; return LAE_OK; }

// how the most recent frame ran: 1 = lookahead on the side stream, 0 = in line (switched off, probed as not concurrent, or degraded after a time-out)
int lae_render_frame_mode(void) {
    std::lock_guard<std::mutex> lk(g_frame_mtx);
    return g_frame.last_mode >= 0 ? g_frame.last_mode : (g_frame_overlap != 0 && !g_frame.degraded ? 1 : 0);
}

// After the caller has synchronised the frame's stream: 0 = the most recent frame completed, 1 = one of its cross-stream waits
// timed out AFTER lae_render_frame had returned (the outputs of that frame are NaN: render it again -- the next call runs in line),
// -1 = no frame yet.  (A time-out the host still sees inside the call is handled there: the frame is rendered again in line.)
int lae_render_frame_last_status(void) {
    std::lock_guard<std::mutex> lk(g_frame_mtx);
    if (!g_frame.ok || g_frame.frame_counter == 0) return -1;
    return g_frame.mirror_h->hung == (uint32_t)g_frame.frame_counter ? 1 : 0;
}

// the most recent side-stream probe: microseconds for 16 hand-overs per candidate (0: configured priority, 1..4: the caller's class;
// < 0 = not probed or not concurrent), up to n values; returns the candidate in use, -1 before the first frame
int lae_render_frame_probe_us(float* us, uint32_t n) {
    std::lock_guard<std::mutex> lk(g_frame_mtx);
    for (uint32_t i = 0; us && i < n; i++) us[i] = i < (uint32_t)FrameHost::MAX_CAND ? g_frame.last_probe_us[i] : -1.f;
    if (!g_frame.ok || g_frame.last_mode < 0) return -1;
    for (int c = 0; c < g_frame.n_cand; c++) if (g_frame.side == g_frame.cand[c]) return c;
    return -1;
}

static int render_frame_once(const float* rays_o, const float* rays_d, uint32_t N, const float* aabb, float min_near,
                     const uint8_t* grid, const uint8_t* edit_grid, float bound, float dt_gamma, uint32_t max_steps,
                     uint32_t C, uint32_t H, const void* table_f16, const int32_t* offsets, const int32_t* offsets_host,
                     uint32_t L, float S,
                     uint32_t base_resolution, uint32_t gridtype, int align_corners, uint32_t interp,
                     const void* sigma_weights, const void* color_weights, float density_scale, float T_thresh,
                     uint32_t max_n_step, uint64_t row_budget, const float* noises, const float* bg_rays, float bg_r, float bg_g,
                     float bg_b, int blend_bg, int scale_depth, float* weights_sum, float* depth, float* image,
                     float* weights_edit, float* depth_edit, void* workspace, uint64_t workspace_bytes, uint32_t* stats_out,
                     void* stream, const bool overlap, bool* gave_up) {
    // one pass of the loop (g_frame_mtx held, arguments checked by lae_render_frame); *gave_up: a cross-stream wait timed out
    // or the device made no progress -- the frame is poisoned (NaN) and both streams are drained
    hipStream_t s = STREAM(stream);
    FrameMirror* mirror_h = g_frame.mirror_h;
    FrameMirror* mirror_d = g_frame.mirror_d;
    const uint64_t frame_id = ++g_frame.frame_counter;
    const uint32_t hung_tag = (uint32_t)frame_id;          // never 0
    uint32_t* hung_dev = reinterpret_cast<uint32_t*>(g_frame.flags + 2);
    const uint32_t spins = frame_wait_spins();
    hipStream_t ls = overlap ? g_frame.side : s;           // stream of the lookahead marcher

    // carve the workspace
    const uint32_t budget = (uint32_t)frame_budget(N, row_budget);
    const uint64_t cap = frame_cap(budget);
    uint8_t* w = reinterpret_cast<uint8_t*>(workspace);
    auto take = [&](uint64_t bytes) { uint8_t* p = w; w += al256(bytes); return p; };
    FrameCtrl* ctrl = reinterpret_cast<FrameCtrl*>(take(256));
    uint32_t* seg_counts[2] = {reinterpret_cast<uint32_t*>(take(4ull * FRAME_SEG_MAX)), reinterpret_cast<uint32_t*>(take(4ull * FRAME_SEG_MAX))};
    uint32_t* blk_counts[2] = {reinterpret_cast<uint32_t*>(take(4ull * FRAME_SEG_MAX)), reinterpret_cast<uint32_t*>(take(4ull * FRAME_SEG_MAX))};
    int32_t* alive = reinterpret_cast<int32_t*>(take(4ull * N));
    int32_t* seg[2] = {reinterpret_cast<int32_t*>(take(4 * frame_seg_elems(N))), reinterpret_cast<int32_t*>(take(4 * frame_seg_elems(N)))};
    RayAcc* acc = reinterpret_cast<RayAcc*>(take(32ull * N));
    float* nears = reinterpret_cast<float*>(take(4ull * N));
    float* fars = reinterpret_cast<float*>(take(4ull * N));
    LookRec rec[2];
    for (int k = 0; k < 2; k++) {
        rec[k].t = reinterpret_cast<float*>(take(4ull * FRAME_LA * N));
        rec[k].e = take((uint64_t)FRAME_LA * N);
        rec[k].cnt = reinterpret_cast<uint32_t*>(take(4ull * N));
        rec[k].tend = reinterpret_cast<float*>(take(4ull * N));
        rec[k].tc = reinterpret_cast<float*>(take(4ull * N));
    }
    float* xyzs = reinterpret_cast<float*>(take(12 * cap));
    float* dirs = reinterpret_cast<float*>(take(12 * cap));
    float* deltas = reinterpret_cast<float*>(take(8 * cap));
    uint8_t* edit_occ = take(cap);
    void* feats = take((uint64_t)L * cap * 4);
    uint32_t* q_counts = reinterpret_cast<uint32_t*>(take(256));            // [0], [1]: task counts, used alternately by consecutive lookaheads; [2], [3]: rays admitted from whole waves
    LookTask* q_tasks = reinterpret_cast<LookTask*>(take(12ull * N));
    static_assert(sizeof(LookTask) == 12, "LookTask layout");
    uint32_t* cmask = reinterpret_cast<uint32_t*>(take(4ull * FRAME_CG * FRAME_CG));
    uint8_t* cdist_buf = take(2ull * FRAME_CG * FRAME_CG * FRAME_CG);   // distances, then octant flags
    static const bool coarse_on = [] { const char* e = getenv("LAE_FRAME_COARSE"); return !e || atoi(e) != 0; }();   // 0: no "nothing ahead" test (A/B)
    const uint8_t* cdist = coarse_on && H >= 4 && (H & (H - 1u)) == 0u && (reinterpret_cast<uintptr_t>(grid) & 3u) == 0u ? cdist_buf : nullptr;   // Morton bytes are 2x2x2 blocks when H is a power of two
    static const bool finish_queue = [] { const char* e = getenv("LAE_FRAME_FINISH_QUEUE"); return !e || atoi(e) != 0; }();   // 0: stragglers finished inside the lane kernel (A/B)
    static const int spec_visits = [] { const char* e = getenv("LAE_FRAME_SPEC"); return e ? atoi(e) : 1; }();   // 0: every visit waits for its own probe (A/B)
    static const int coop_max_env = [] { const char* e = getenv("LAE_FRAME_COOP_MAX"); return e ? atoi(e) : -1; }();
    const int coop_max = coop_max_env >= 0 ? coop_max_env : (finish_queue ? FRAME_QUEUE_MAX : FRAME_COOP_MAX);
    const int max_rounds = FRAME_MAX_ROUNDS;                // (sweep 4 / 6 / 8 / 12 closed in round 4: lanes keep walking, DESIGN_LOG)
    // the first walk of a frame is another regime: every ray that hits anything first crosses empty space, a wave's lanes finish
    // at very different times and a low bound sends tens of thousands of rays to the one-wave-per-ray kernel
    const int coop_max0 = 0;                                // every lane of the first walk finishes in the lane kernel (sweep -1 / 0 / 2 / 4 / 8 / 32: 10.80 / 10.70 / 10.81 / 10.79 / 10.82 / 11.09 ms at 800x800; closed)
    static const uint32_t admit_cap = [] { const char* e = getenv("LAE_FRAME_ADMIT_CAP"); return e ? (uint32_t)atoi(e) : 4096u; }();   // 0: never (A/B)
    static const int admit_round = [] { const char* e = getenv("LAE_FRAME_ADMIT_ROUND"); return e ? atoi(e) : 2; }();
    uint32_t look_no = 0;

    const MarchCfg cfg = make_cfg(bound, dt_gamma, max_steps, C, H);
    const float in_shift = bound, in_scale = 1.0f / (2.0f * bound);        // grid.py:149 (torch multiplies by the fp32 reciprocal)
    // lookahead of iteration `phase` (-1: before the loop): rays = the survivor lists `sg` of iteration phase - 1, records
    // read from `in`, written to `out`; `waves` sizes the launch (one wave per 64 list entries), n_bound the finishing kernel
    auto lookahead = [&](int phase, const FrameCtrl* pv, const FrameSegs& sg, const LookRec& in, const LookRec& out, uint32_t waves,
                         uint32_t n_bound, hipStream_t q) {
        const uint32_t blocks = lae::cdiv(waves, FRAME_BLOCK / 64);
        uint32_t* qc = q_counts + (look_no & 1u);
        uint32_t* qc_next = q_counts + ((look_no + 1u) & 1u);
        look_no++;
        LookTask* qt = finish_queue ? q_tasks : nullptr;
        const uint32_t fin_blocks = std::min(FRAME_FINISH_BLOCKS, std::max(1u, lae::cdiv(n_bound, FRAME_BLOCK / 64)));
        if (edit_grid) {
            k_frame_lookahead<true><<<blocks, FRAME_BLOCK, 0, q>>>(phase, pv, sg, N, budget, max_steps, max_n_step, in, out, rays_o, rays_d, fars,
                                                                 cfg, grid, edit_grid, noises, qc, qt, spec_visits, phase < 0 && coop_max0 >= 0 ? coop_max0 : coop_max, max_rounds, cdist, phase < 0 ? 0u : admit_cap, admit_round, hung_dev, hung_tag);
            if (qt) k_frame_lookahead_finish<true><<<fin_blocks, FRAME_BLOCK, 0, q>>>(max_n_step, qc, qc_next, qt, out, rays_o, rays_d, fars, cfg, grid, edit_grid, cdist);
        } else {
            k_frame_lookahead<false><<<blocks, FRAME_BLOCK, 0, q>>>(phase, pv, sg, N, budget, max_steps, max_n_step, in, out, rays_o, rays_d, fars,
                                                                  cfg, grid, nullptr, noises, qc, qt, spec_visits, phase < 0 && coop_max0 >= 0 ? coop_max0 : coop_max, max_rounds, cdist, phase < 0 ? 0u : admit_cap, admit_round, hung_dev, hung_tag);
            if (qt) k_frame_lookahead_finish<false><<<fin_blocks, FRAME_BLOCK, 0, q>>>(max_n_step, qc, qc_next, qt, out, rays_o, rays_d, fars, cfg, grid, nullptr, cdist);
        }
    };
    k_near_far<<<lae::cdiv(N, 256), 256, 0, s>>>(rays_o, rays_d, aabb, N, min_near, nears, fars);
    k_frame_init<<<lae::cdiv(N, 256), 256, 0, s>>>(ctrl, N, nears, acc, rec[0].tc, q_counts, cdist ? cmask : nullptr);
    if (cdist) {                                          // the bitfield is the caller's and may have changed since the last frame
        k_frame_coarse_mark<<<std::min(64u, lae::cdiv(C * (H * H * H / 8u), 4096u)), 1024, 0, s>>>(grid, C, H, bound, cmask);
        k_frame_coarse_dist<<<2, 1024, 0, s>>>(cmask, cdist_buf);
    }
    const FrameSegs no_segs{nullptr, nullptr, nullptr, 0u, 0u};
    lookahead(-1, nullptr, no_segs, rec[0], rec[0], lae::cdiv(N, 64), N, s);
    // Two chains per iteration i (records double-buffered: emit(i) and lookahead(i) read buffer i & 1 and the survivor
    // segments of head(i-1), lookahead(i) writes buffer (i+1) & 1):
    //   caller's stream:  wait(look = i)  -> emit(i) [go = i+1, at its start or its end] -> encoder(i) -> head + compositing(i)
    //   side stream:      wait(go = i+1)  -> lookahead(i) -> its finishing kernel -> signal(look = i+1)
    // go at the START of the emit kernel: lookahead(i) needs nothing emit(i) writes, only that head(i-1) is complete, which the
    // emit kernel running proves.  (LAE_FRAME_LOOK_EARLY=0: go when the emit kernel's last workgroup is done, so that the
    // lookahead's loads do not compete with the emit kernel on the caller's critical path -- measured 11.83 / 69.8 / 12.47 ms
    // against 11.40 / 69.1 / 11.73 ms for the 800x800 frame / the 1080p frame / one rank's shard of it: the later start costs
    // the lookahead chain more than the quieter emit kernel gains.)
    // rows of n_step >= 3 iterations through a wave-private LDS image (k_frame_emit); the host knows a lagging bound of n_alive,
    // so whether an iteration can have n_step >= 3 at all (budget / bound_alive >= 3) decides if the launch asks for the LDS
    static const int grid_tail = [] { const char* e = getenv("LAE_FRAME_GRID_TAIL"); return e ? atoi(e) : 1; }();   // 0: one encoder launch sized for the worst case (A/B)
    const int emit_lds_min = 3;                             // smallest n_step that takes the LDS path (sweep 3 / 5 / 7 / 8 / never closed in round 4)
    static const int go_early = [] { const char* e = getenv("LAE_FRAME_LOOK_EARLY"); return e ? (atoi(e) != 0) : 1; }();
    unsigned long long* flag_go = g_frame.flags;
    unsigned long long* flag_look = g_frame.flags + 1;
    const unsigned long long fbase = (unsigned long long)frame_id << 32;
    volatile uint32_t* hung_d = &mirror_d->hung;
    uint32_t bound_alive = N, seen_iter = 0;
    bool done = false;
    int rc = LAE_OK;
    static const uint32_t LAG = [] { const char* e = getenv("LAE_FRAME_LAG"); return e ? (uint32_t)std::max(atoi(e), 1) : 2u; }();   // iterations the host may run ahead of the mirror.  Launches are sized from the mirror's n_alive: the fresher it is the fewer
                                                           // workgroups find nothing to do (1 / 2 / 3 / 4 / 6: 800x800 9.99 / 10.01 / 10.03 / 10.15 / 10.23 ms, 1080p 56.4 / 56.4 / 57.4 / 57.4 / 58.4, shard 8.45 / 8.44 / 8.56 / 8.62 / 8.75;
                                                           // the host needs ~45 us to launch an iteration that takes the device 130+: two iterations of slack are enough on this pool)
    uint32_t it = 0, nu_prev = 0, R_prev = 0;             // survivor segments (= waves of the previous k_frame_head) and their stride
    auto join_side = [&]() {                               // every exit path: the caller's stream owns the workspace again
        if (overlap && it > 0) k_frame_wait<<<1, 64, 0, s>>>(flag_look, fbase | it, hung_dev, hung_d, hung_tag, spins);
    };
    auto abort_frame = [&]() {                             // a launch failed: release whoever polls for work that will never come
        if (overlap) {
            k_frame_signal<<<1, 1, 0, s>>>(flag_go, fbase | 0xffffffffull);
            k_frame_signal<<<1, 1, 0, ls>>>(flag_look, fbase | 0xffffffffull);
            (void)hipStreamSynchronize(ls);
        }
    };
    auto give_up = [&]() {                                 // the handshake made no progress: drain both streams, the caller renders again in line
        abort_frame();
        (void)hipStreamSynchronize(s);
        *gave_up = true;
        lae::set_last_error_str("render_frame: device made no progress");
        return LAE_ELAUNCH;
    };
    auto poll = [&]() {
        if (mirror_h->tag != frame_id) return;
        const uint32_t iters = mirror_h->iters, na = mirror_h->n_alive, dn = mirror_h->done;
        if (iters > seen_iter) seen_iter = iters;
        if (na < bound_alive) bound_alive = na;
        // the four mirror words are separate stores: n_alive == 0 may be visible before its done flag (the device sets
        // n_alive to 0 only together with done), and a zero bound would size the next launches to zero workgroups
        if (dn || na == 0) done = true;
    };
#ifdef LAE_GRID_STAMPS
    static const uint32_t stop_after = [] { const char* e = getenv("LAE_FRAME_STOP_AFTER"); return e ? (uint32_t)atoi(e) : 0u; }();   // probe builds: the frame ends after that many iterations (tools/frame_grid_spans.py)
#endif
    for (; it <= max_steps && !done; it++) {
#ifdef LAE_GRID_STAMPS
        if (stop_after && it >= stop_after) break;
#endif
        poll();
        if (done) break;
        if (it >= seen_iter + LAG) {                       // do not run further ahead than LAG iterations
            const auto t0 = std::chrono::steady_clock::now();
            while (it >= seen_iter + LAG && !done) {
                poll();
                if (mirror_h->hung == hung_tag || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return give_up();
            }
            if (done) break;
        }
        const FrameCtrl* prev = ctrl + (it & 1u);
        FrameCtrl* cur = ctrl + ((it + 1u) & 1u);
        const uint32_t p = it & 1u;                        // segments / counts written by this iteration's head: [p]; read: [p ^ 1]; records read: [p]
        const FrameSegs sg{seg_counts[p ^ 1u], blk_counts[p ^ 1u], seg[p ^ 1u], nu_prev, R_prev};
        const uint32_t list_waves = nu_prev ? nu_prev * (R_prev / 64u) : lae::cdiv(N, 64);
        const uint32_t rows_bound = (uint32_t)std::min<uint64_t>((uint64_t)budget, (uint64_t)max_n_step * bound_alive);
        const uint32_t rows_launch = (uint32_t)std::min<uint64_t>(frame_padded_rows(rows_bound), cap);
        // side chain: lookahead for the NEXT iteration's samples
        if (overlap) k_frame_wait<<<1, 64, 0, ls>>>(flag_go, fbase | (it + 1u), hung_dev, hung_d, hung_tag, spins);
        lookahead((int)it, prev, sg, rec[p], rec[p ^ 1u], list_waves, bound_alive, ls);
        if (overlap) k_frame_signal<<<1, 1, 0, ls>>>(flag_look, fbase | (it + 1u));
        // caller's chain: the samples of this iteration come from the previous lookahead
        join_side();
        const uint32_t emit_blocks = lae::cdiv(list_waves, FRAME_BLOCK / 64);
        const int emit_lds = emit_lds_min >= 2 && max_n_step >= (uint32_t)emit_lds_min && (uint64_t)budget >= (uint64_t)emit_lds_min * bound_alive ? emit_lds_min : 0;
        // the image is sized for one n_step more than the bound implies (the bound lags a few iterations): 5 KB per wave at
        // n_step = 3, 13 KB at 8 -- at the full size only three workgroups fit a CU and the kernel ran in two rounds
        const uint32_t n_lb = (uint32_t)std::min<uint64_t>((uint64_t)budget / std::max(bound_alive, 1u), (uint64_t)max_n_step);
        const uint32_t img_rows = emit_img_rows(std::min(std::max(n_lb, 1u) + 1u, max_n_step));
        const uint32_t emit_lds_bytes = emit_lds ? (FRAME_BLOCK / 64) * emit_img_floats(img_rows) * 4u : 0u;
        static const int emit8_on = [] { const char* e = getenv("LAE_FRAME_EMIT8"); return e ? atoi(e) : 1; }();   // 0: the general emit kernel for n_step = 8 too (A/B)
        if (emit8_on && emit_lds && n_lb >= 8u && edit_grid)
            k_frame_emit8<true><<<emit_blocks, FRAME_BLOCK, 0, s>>>(prev, cur, sg, N, budget, max_steps, max_n_step, alive, rec[p], rays_o, rays_d, cfg,
                                                                 xyzs, dirs, deltas, edit_occ, noises, mirror_d, frame_id,
                                                                 overlap ? flag_go : nullptr, fbase | (it + 1u), go_early);
        else if (emit8_on && emit_lds && n_lb >= 8u)
            k_frame_emit8<false><<<emit_blocks, FRAME_BLOCK, 0, s>>>(prev, cur, sg, N, budget, max_steps, max_n_step, alive, rec[p], rays_o, rays_d, cfg,
                                                                  xyzs, dirs, deltas, nullptr, noises, mirror_d, frame_id,
                                                                  overlap ? flag_go : nullptr, fbase | (it + 1u), go_early);
        else if (edit_grid)
            k_frame_emit<true><<<emit_blocks, FRAME_BLOCK, emit_lds_bytes, s>>>(prev, cur, sg, N, budget, max_steps, max_n_step, alive, rec[p], rays_o, rays_d, cfg,
                                                                xyzs, dirs, deltas, edit_occ, noises, mirror_d, frame_id,
                                                                overlap ? flag_go : nullptr, fbase | (it + 1u), go_early, emit_lds, img_rows);
        else
            k_frame_emit<false><<<emit_blocks, FRAME_BLOCK, emit_lds_bytes, s>>>(prev, cur, sg, N, budget, max_steps, max_n_step, alive, rec[p], rays_o, rays_d, cfg,
                                                                 xyzs, dirs, deltas, nullptr, noises, mirror_d, frame_id,
                                                                 overlap ? flag_go : nullptr, fbase | (it + 1u), go_early, emit_lds, img_rows);
        // rows the host EXPECTS: its bound of the rays alive x the n_step that bound implies (a second, small launch covers the
        // rows beyond, which exist only in the iterations after n_step rose on the device)
        const uint32_t rows_likely = grid_tail ? (uint32_t)std::min<uint64_t>(frame_padded_rows((uint64_t)bound_alive * std::max(n_lb, 1u)), rows_launch) : rows_launch;
        rc = lae::grid_forward_frame(xyzs, table_f16, offsets, feats, (uint32_t)cap, rows_launch, &cur->n_rows, L, S, base_resolution,
                                     gridtype, align_corners, interp, in_shift, in_scale, s, offsets_host, rows_likely);
        // head + compositing: one wave per run of 64-row groups; the survivors of wave u go, in order, to segment u of
        // stride R (> the rays a wave can own: bound_alive / units + 64 / units + 64) with their count
        if (rc == LAE_OK) {
            lae::FrameHeadArgs fa;
            fa.cur = cur; fa.alive = alive; fa.deltas = deltas; fa.edit_occ = edit_grid ? edit_occ : nullptr; fa.acc = acc;
            fa.seg_next = seg[p]; fa.seg_counts_next = seg_counts[p]; fa.blk_counts_next = blk_counts[p]; fa.T_thresh = T_thresh;
            const uint32_t head_blocks = std::max(1u, std::min(lae::cdiv(rows_launch, 64u * lae::FRAME_HEAD_WAVES), lae::frame_head_max_blocks()));
            const uint32_t units = head_blocks * lae::FRAME_HEAD_WAVES;
            fa.R = ((bound_alive / units + 66u + 63u) / 64u) * 64u;
            if (units > FRAME_SEG_MAX || (uint64_t)units * fa.R > frame_seg_elems(N)) {
                lae::set_last_error_str("render_frame: survivor segments do not fit their buffers");
                rc = LAE_EINVAL;
            } else {
                rc = lae::nerf_head_composite_frame(feats, dirs, sigma_weights, color_weights, (uint32_t)cap, density_scale, fa,
                                                    edit_grid != nullptr, head_blocks, s);
                nu_prev = units; R_prev = fa.R;
            }
        }
        if (rc == LAE_OK) rc = lae::check_launch("render_frame");
        if (rc) { abort_frame(); return rc; }
    }
    join_side();
    k_frame_finish<<<lae::cdiv(N, 256), 256, 0, s>>>(N, nears, fars, acc, weights_sum, depth, image, edit_grid ? weights_edit : nullptr,
                                                     edit_grid ? depth_edit : nullptr, bg_rays, bg_r, bg_g, bg_b, blend_bg, scale_depth,
                                                     hung_dev, hung_tag);
    rc = lae::check_launch("render_frame(finish)");
    if (rc) return rc;
    if (mirror_h->hung == hung_tag) return give_up();      // a wait gave up behind the host's back (the finish kernel wrote NaN)
    if (stats_out) {
        // the loop ends either on `done` (mirror holds the final state) or after max_steps + 1 launched iterations
        const auto t0 = std::chrono::steady_clock::now();
        while (!(mirror_h->tag == frame_id && mirror_h->done)) {     // the done flag is the last word the device writes
            if (mirror_h->hung == hung_tag) return give_up();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) break;
        }
        stats_out[0] = mirror_h->iters; stats_out[1] = mirror_h->total_rows; stats_out[2] = it;
    }
    return LAE_OK;
}

int lae_render_frame(const float* rays_o, const float* rays_d, uint32_t N, const float* aabb, float min_near,
                     const uint8_t* grid, const uint8_t* edit_grid, float bound, float dt_gamma, uint32_t max_steps,
                     uint32_t C, uint32_t H, const void* table_f16, const int32_t* offsets, const int32_t* offsets_host,
                     uint32_t L, float S,
                     uint32_t base_resolution, uint32_t gridtype, int align_corners, uint32_t interp,
                     const void* sigma_weights, const void* color_weights, float density_scale, float T_thresh,
                     uint32_t max_n_step, uint64_t row_budget, const float* noises, const float* bg_rays, float bg_r, float bg_g,
                     float bg_b, int blend_bg, int scale_depth, float* weights_sum, float* depth, float* image,
                     float* weights_edit, float* depth_edit, void* workspace, uint64_t workspace_bytes, uint32_t* stats_out,
                     void* stream) {
    if (N == 0) return LAE_OK;
    if (!rays_o || !rays_d || !aabb || !grid || !table_f16 || !offsets || !sigma_weights || !color_weights || !weights_sum ||
        !depth || !image || !workspace)
        return LAE_ENULL;
    if (edit_grid && (!weights_edit || !depth_edit)) return LAE_ENULL;
    if (C == 0 || C > 8 || H == 0 || H > 1024 || max_steps == 0 || max_n_step == 0 || max_n_step > FRAME_LA || L != 16) return LAE_EINVAL;
    if (workspace_bytes < lae_render_frame_workspace_bytes(N, L, row_budget)) return LAE_EINVAL;
    hipStream_t s = STREAM(stream);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
        lae::set_last_error_str("render_frame: the frame loop adapts its launches to the device state and cannot be stream-captured");
        return LAE_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_frame_mtx);
    if (!g_frame.init()) { lae::set_last_error_str("render_frame: could not create the pinned mirror / side stream / events"); return LAE_ELAUNCH; }
    // One mirror / side stream / flag pair per process (one process per GPU): frames on the SAME stream are ordered by the
    // stream itself; a frame on another stream first waits for the previous frame's queued tail.
    if (g_frame.have_last && g_frame.last_stream != s) (void)hipStreamSynchronize(g_frame.last_stream);
    g_frame.last_stream = s; g_frame.have_last = true;
    // a wait of the PREVIOUS frame gave up after its call had returned (its outputs hold NaN, k_frame_finish): never overlap again
    if (!g_frame.degraded && g_frame.frame_counter && g_frame.mirror_h->hung == (uint32_t)g_frame.frame_counter) {
        g_frame.degraded = true;
        frame_warn_once("a cross-stream wait of the previous frame timed out (that frame's outputs are NaN)");
    }
    // Lookahead beside the encoder / head kernels (a side stream, ordered by two polled words) needs the two streams to run
    // CONCURRENTLY; when they cannot -- probed once per caller stream, or found out the hard way by a wait that gave up --
    // the same kernels run in line on the caller's stream: the image has the same bits, the frame is slower.
    static const int forced = [] { const char* e = getenv("LAE_FRAME_OVERLAP"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();   // 1: skip the probe (tests of the give-up path), 0: in line
    bool overlap = g_frame_overlap != 0 && !g_frame.degraded && forced != 0;
    if (overlap && forced < 0) {
        hipStream_t pick = frame_pick_side(s);
        if (pick) g_frame.side = pick;
        else {
            overlap = false;
            frame_warn_once("the caller's stream and the side stream do not run concurrently (one hardware queue, serialised dispatch or a counter-collecting profiler)");
        }
    }
    bool gave_up = false;
#define LAE_FRAME_ARGS rays_o, rays_d, N, aabb, min_near, grid, edit_grid, bound, dt_gamma, max_steps, C, H, table_f16, offsets, offsets_host, L, S, \
        base_resolution, gridtype, align_corners, interp, sigma_weights, color_weights, density_scale, T_thresh, max_n_step, row_budget, noises,      \
        bg_rays, bg_r, bg_g, bg_b, blend_bg, scale_depth, weights_sum, depth, image, weights_edit, depth_edit, workspace, workspace_bytes,           \
        stats_out, stream
    g_frame.last_mode = overlap ? 1 : 0;
    int rc = render_frame_once(LAE_FRAME_ARGS, overlap, &gave_up);
    if (gave_up && overlap) {                              // degrade instead of failing: once, in line, and remember it for the process
        g_frame.degraded = true;
        frame_warn_once("a cross-stream wait timed out");
        gave_up = false;
        g_frame.last_mode = 0;
        rc = render_frame_once(LAE_FRAME_ARGS, false, &gave_up);
    }
#undef LAE_FRAME_ARGS
    return rc;
}

}  // extern "C"

#ifdef LAE_FRAME_STAMPS
extern "C" __attribute__((visibility("default"))) int lae_debug_look_stamps(void* out, size_t bytes, int phase) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (phase >= 0) return hipMemcpyToSymbol(HIP_SYMBOL(g_look_phase), &phase, sizeof(int)) == hipSuccess ? 0 : 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_look_stamps), bytes) == hipSuccess ? 0 : 1;
}
#endif
