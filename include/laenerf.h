/*
 * laenerf.h -- C ABI of the MI355X-native LAENeRF / torch-ngp hot path.
 *
 * One symbol per backend function of the reference's four pybind11 modules
 * (_raymarching, _gridencoder, _shencoder, _ffmlp).  Every entry point
 *   - takes raw DEVICE pointers + sizes + scalars (no torch types),
 *   - borrows the pointers for the duration of the call only (the caller owns
 *     and pre-allocates every output / workspace, exactly like the reference's
 *     autograd.Functions do),
 *   - enqueues its kernels on `stream` (a hipStream_t passed as void*; NULL =
 *     the legacy default stream, which is what the reference launched on),
 *   - never synchronises the host and never frees or reallocates memory the caller
 *     passed in.  LIBRARY-OWNED WORKSPACES (the counterpart of the reference's
 *     process-global split-K streams + CUTLASS workspace, ffmlp.cu:711-740,
 *     cutlass_matmul.h:335-352): the fp32 dW slabs of the MLP backward, the
 *     level-major staging of the [B, L*C] grid-encoder layouts and the bin queues of
 *     the grid backward live in grow-only device buffers, one set per device.
 *     WARM-UP CONTRACT: the first call at a new largest size hipMalloc()s (a host-
 *     side, possibly blocking, call); steady state allocates nothing.  A buffer
 *     that is outgrown is retired, NOT freed -- kernels in flight and captured HIP
 *     graphs keep using the address they were given -- until lae_free_workspaces().
 *     Growth is refused (LAE_ELAUNCH, message in lae_last_error) while `stream` is
 *     being captured: run the call once eagerly at its largest size before
 *     capturing.  The library acts on the CURRENT device of the calling thread;
 *     the caller makes the device of its pointers current (hipSetDevice) first,
 *   - returns LAE_OK (0) or a negative LAE_E* code; the Python shim turns a
 *     non-zero code into RuntimeError (the reference threw c10::Error /
 *     std::runtime_error for the same conditions).
 *
 * All tensors are dense row-major ("contiguous") exactly as the reference
 * requires (CHECK_CONTIGUOUS, gridencoder.cu:455-459).
 *
 * Reference interface replaced, per symbol: see the comment above each
 * prototype (paths relative to the reference checkout).
 */
#ifndef LAENERF_H
#define LAENERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported from liblaenerf_hip.so (the library is built with -fvisibility=hidden) */
#if defined(__GNUC__) || defined(__clang__)
#define LAE_API __attribute__((visibility("default")))
#else
#define LAE_API
#endif

#define LAE_OK 0
#define LAE_EINVAL (-1)     /* unsupported template parameter / bad argument  */
#define LAE_ELAUNCH (-2)    /* hipGetLastError() != hipSuccess after a launch */
#define LAE_ENULL (-3)      /* required pointer is NULL                       */

/* element type of the hash table / encoder outputs / encoder grads */
#define LAE_F32 0
#define LAE_F16 1

/* grid types / interpolation (gridencoder/grid.py:14-22) */
#define LAE_GRID_HASH 0
#define LAE_GRID_TILED 1
#define LAE_INTERP_LINEAR 0
#define LAE_INTERP_SMOOTHSTEP 1

/* ffmlp activations (ffmlp/ffmlp.py:89-96, ffmlp/src/utils.h:29-37) */
#define LAE_ACT_RELU 0
#define LAE_ACT_EXPONENTIAL 1
#define LAE_ACT_SINE 2
#define LAE_ACT_SIGMOID 3
#define LAE_ACT_SQUAREPLUS 4
#define LAE_ACT_SOFTPLUS 5
#define LAE_ACT_NONE 6

/* library identification: returns the static string "laenerf-hip gfx950 " LAE_ABI_TAG.  The tag changes whenever a
 * signature of this header changes incompatibly (abi2: round 2 added pointer arguments in the middle of the optimizer /
 * grid-backward / frame entry points; abi3: round 3, optimizer state words and the compositing step; abi4: round 4, lae_ffmlp_set_mode values 2 and 16-18 removed; abi5: round 5, lae_render_frame_mode, frame-loop degrade path; abi6: round 6, lae_render_frame_last_status, lae_ffmlp_forward leaves forward_buffer untouched where the backward recomputes).  A binding compares
 * it with the tag it was written against BEFORE the first call: a stale .so used through newer prototypes would misalign
 * arguments silently (laenerf_amd/_lib.py does, and rebuilds or raises). */
#define LAE_ABI_TAG "abi6"
LAE_API const char* lae_version(void);
/* last HIP error string recorded by a failed launch in this thread (or "") */
LAE_API const char* lae_last_error(void);

/* ------------------------------------------------------------------ */
/* _raymarching  (raymarching/src/raymarching.h:7-20, bindings.cpp:5-20) */
/* ------------------------------------------------------------------ */

/* raymarching.cu:148-156  near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars) */
LAE_API int lae_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                           uint32_t N, float min_near, float* nears, float* fars, void* stream);

/* nerf/utils.py:61-153  get_rays(poses, intrinsics, H, W, N, ...): MI355X-native -- the ray generation of the data loader
 * (meshgrid + gather + stack + normalise + matmul, ~15 torch launches) as one kernel.  poses [B,4,4] cam2world, row-major;
 * inds [B,N] int64 flat pixel indices h*W+w with batch stride `inds_batch_stride` elements (0: one index row shared by all
 * poses, the reference's `inds.expand([B, N])`), or NULL for every pixel in order (then N must equal H*W).  perturb != 0
 * subtracts (off_x, off_y) from the pixel centre (`perturb_ray_dirs`, :133-136).  rays_o, rays_d [B,N,3].  aabb != NULL
 * additionally writes the near_far_from_aabb interval (raymarching.cu:91-145) of every ray to nears, fars [B,N]. */
LAE_API int lae_get_rays(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                 const int64_t* inds, uint64_t inds_batch_stride, uint32_t N, int perturb, float off_x, float off_y,
                 float* rays_o, float* rays_d, const float* aabb, float min_near, float* nears, float* fars, void* stream);

/* raymarching.cu:201-209  sph_from_ray(rays_o, rays_d, radius, N, coords[N,2]) */
LAE_API int lae_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N,
                     float* coords, void* stream);

/* raymarching.cu:229-232  morton3D(coords[N,3] i32, N, indices[N] i32) */
LAE_API int lae_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream);

/* raymarching.cu:257-260  morton3D_invert(indices[N], N, coords[N,3]) */
LAE_API int lae_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream);

/* raymarching.cu:292-300  packbits(grid[8N] f32, N bytes, thresh, bitfield[N] u8) */
LAE_API int lae_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield,
                 void* stream);

/* raymarching.cu:482-490  march_rays_train(...)
 * rays[N,3] = (ray id, offset, count).  Rows are written in RAY-ID order with
 * offsets = exclusive prefix sum of counts (the reference's atomicAdd
 * reservation order is non-deterministic; sequential execution of the
 * reference gives exactly this order).  counter[0] += sum(count),
 * counter[1] += N, like the reference's atomicAdds on a pre-zeroed counter.
 * `scratch` : caller-provided device workspace of lae_march_rays_train_scratch_bytes(N)
 * bytes (may be NULL only when N == 0): per-ray counts and offsets and, for N <= 2^18, the chunk records
 * the counting pass leaves for the emitting pass (292 B per ray).
 * Sample rows no ray owns, [rows_end, M), are zero-filled by the kernel itself (the reference relies on the caller's
 * torch.zeros, raymarching.py:207-209); rows_end is also stored to rows_end_out (device uint32, may be NULL) for
 * lae_composite_rays_train_backward_blend. */
LAE_API uint64_t lae_march_rays_train_scratch_bytes(uint32_t N);
LAE_API int lae_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid,
                         float bound, float dt_gamma, uint32_t max_steps,
                         uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float* nears, const float* fars,
                         float* xyzs, float* dirs, float* deltas,
                         int32_t* rays, int32_t* counter, const float* noises,
                         void* scratch, uint32_t* rows_end_out, void* stream);

/* raymarching.cu:580-588 */
LAE_API int lae_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                     const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                     float* weights_sum, float* depth, float* image, void* stream);

/* raymarching.cu:685-693 */
LAE_API int lae_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                      const float* sigmas, const float* rgbs, const float* deltas,
                                      const int32_t* rays, const float* weights_sum,
                                      const float* image, uint32_t M, uint32_t N, float T_thresh,
                                      float* grad_sigmas, float* grad_rgbs, void* stream);

/* MI355X-native: composite + the post-ops run_cuda applies to its result (nerf/renderer.py:321, 325) in one kernel:
 *   image_out = image + (1 - weights_sum) * bg,  depth_out = clamp(depth - nears, min=0) / (fars - nears).
 * bg = bg_rays [N,3] when non-NULL, else the constant (bg_r, bg_g, bg_b).  weights_sum / depth / image receive the
 * un-blended values (saved for the backward). */
LAE_API int lae_composite_rays_train_forward_blend(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                           uint32_t M, uint32_t N, float T_thresh, const float* nears, const float* fars,
                                           const float* bg_rays, float bg_r, float bg_g, float bg_b, float* weights_sum,
                                           float* depth, float* image, float* depth_out, float* image_out, void* stream);
/* MI355X-native: lae_composite_rays_train_forward_blend + lae_mse_loss_forward (target [N,3], scale as there) +
 * lae_composite_rays_train_backward_blend (grad_weights_sum = NULL, upstream gradient 1) as ONE launch -- d loss / d pixel of a
 * ray needs that ray's pixel only -- plus a one-block launch that adds the workgroups' squared-error sums in a fixed order into
 * loss_out[0] = MSE * scale, loss_out[1] = MSE.  Same arithmetic as the three calls.  grad_image [N,3], grad_sigmas [M],
 * grad_rgbs [M,3] are outputs (every row written); partials: cdiv(N, 4) floats of scratch. */
LAE_API int lae_composite_rays_train_step(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M,
                                  uint32_t N, float T_thresh, const float* nears, const float* fars, const float* bg_rays, float bg_r,
                                  float bg_g, float bg_b, const uint32_t* rows_end, const float* target, const float* scale,
                                  float* weights_sum, float* depth, float* image, float* depth_out, float* image_out,
                                  float* grad_image, float* grad_sigmas, float* grad_rgbs, float* loss_out, float* partials,
                                  int defer_loss, void* stream);
/* defer_loss != 0 (round 3): the one-block launch is left out and loss_out[0..1] is set to NaN; the sum is done later by
 * lae_loss_finish(partials, cdiv(N, 4), 3 * N, scale, loss_out) or taken along by lae_nerf_head_backward (its loss_*
 * arguments) -- the value feeds nothing on the device, so it need not sit on the step's critical path. */
LAE_API int lae_loss_finish(const float* partials, uint32_t n_part, uint32_t n_elem, const float* scale, float* loss_out, void* stream);

/* Backward of the above w.r.t. (weights_sum, image_out): grad_ws_eff = grad_ws - sum_c grad_image_c * bg_c.  Writes EVERY
 * row of grad_sigmas / grad_rgbs in [0, M) (zeros after the early stop and in [rows_end, M)), so they need no
 * pre-zeroing; requires the contiguous ray-id-order sample layout of lae_march_rays_train. */
LAE_API int lae_composite_rays_train_backward_blend(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                            const float* rgbs, const float* deltas, const int32_t* rays,
                                            const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                            float T_thresh, const float* bg_rays, float bg_r, float bg_g, float bg_b,
                                            const uint32_t* rows_end, float* grad_sigmas, float* grad_rgbs, void* stream);
/* same, for a criterion fused in front of it: grad_weights_sum may be NULL (zero) and every incoming gradient is
 * multiplied by the device scalar *grad_scale when grad_scale != NULL (the upstream d(loss), so no elementwise
 * multiplication / zero-fill kernels are needed between the loss and this call). */
LAE_API int lae_composite_rays_train_backward_blend_ex(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                               const float* rgbs, const float* deltas, const int32_t* rays,
                                               const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                               float T_thresh, const float* bg_rays, float bg_r, float bg_g, float bg_b,
                                               const uint32_t* rows_end, const float* grad_scale, float* grad_sigmas,
                                               float* grad_rgbs, void* stream);

/* raymarching.cu:929-936 */
LAE_API int lae_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
                   const float* rays_t, const float* rays_o, const float* rays_d,
                   float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                   const uint8_t* grid, const float* nears, const float* fars,
                   float* xyzs, float* dirs, float* deltas, const float* noises, void* stream);

/* raymarching.cu:938-945  (edit_occ is a torch.bool tensor = 1 byte / sample) */
LAE_API int lae_march_rays_distill(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
                           const float* rays_t, const float* rays_o, const float* rays_d,
                           float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                           const uint8_t* grid, const uint8_t* edit_grid,
                           const float* nears, const float* fars,
                           float* xyzs, float* dirs, float* deltas, uint8_t* edit_occ,
                           const float* noises, void* stream);

/* raymarching.cu:1145-1151 */
LAE_API int lae_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                       int32_t* rays_alive, float* rays_t,
                       const float* sigmas, const float* rgbs, const float* deltas,
                       float* weights_sum, float* depth, float* image, void* stream);

/* raymarching.cu:1153-1159 */
LAE_API int lae_composite_rays_distill(uint32_t n_alive, uint32_t n_step, float T_thresh,
                               int32_t* rays_alive, float* rays_t,
                               const float* sigmas, const float* rgbs, const float* deltas,
                               float* weights_sum, float* weights_edit_sum,
                               float* depth, float* depth_edit,
                               const uint8_t* edit_occ, float* image, void* stream);

/* MI355X-native extension: the whole inference loop of NeRFRenderer.run_cuda (nerf/renderer.py:335-387) -- and of
 * run_cuda_distill (:394-480) when edit_grid != NULL -- as ONE call.  near_far_from_aabb, then per iteration
 * march_rays -> hash-grid encode -> fused sigma/colour head -> composite_rays -> alive-list compaction, with
 * n_alive / n_step / step kept in device memory and advanced on the device exactly like the Python
 * (`n_step = max(min(row_budget // n_alive, max_n_step), 1)`; the reference has row_budget = N (pass 0) and
 * max_n_step = 8; a larger row_budget trades buffer memory for fewer, larger iterations -- per-ray sample sequences
 * are unchanged, only the float rounding of rays_t between iterations can differ; loop ends when no ray is alive or
 * step >= max_steps), then `image + (1 - weights_sum) * bg` (blend_bg) and `clamp(depth - nears, 0) / (fars - nears)`
 * (scale_depth).  The host never waits for the device inside the loop: it sizes launches from a lagging upper bound of
 * n_alive mirrored into pinned memory.  Not stream-capturable (LAE_EINVAL while capturing).
 *   table_f16 [sum level sizes, 2] fp16, offsets [L+1] int32 (device), offsets_host = the same L+1 ints in HOST memory
 *   or NULL (only the level -> XCD balance of the encoder depends on it), L = 16, S = log2(per_level_scale), base_resolution;
 *   sigma_weights / color_weights: FFMLP flat fp16 images of the 32->64->64->16 and 32->64->64->64->16 nets;
 *   noises [N] or NULL (perturb, applied in the first iteration only, renderer.py:365); bg_rays [N,3] or NULL (then
 *   bg_r/g/b); outputs weights_sum, depth [N], image [N,3] fp32 (+ weights_edit, depth_edit [N] when edit_grid);
 *   workspace: lae_render_frame_workspace_bytes(N, L, row_budget) bytes of device memory owned by the caller;
 *   stats_out (host, may be NULL; makes the call wait for the loop's last iteration): [iterations, rows through the
 *   network, iterations launched]. */
LAE_API uint64_t lae_render_frame_workspace_bytes(uint32_t N, uint32_t L, uint64_t row_budget);
/* 1 (default) runs the lookahead marcher on a library-owned side stream beside the encoder / MLP kernels, 0 runs it
 * in-line on the caller's stream (same image bit for bit).  CONCURRENCY REQUIREMENT of the overlapped form: the two streams
 * are ordered by polled words in device memory, not by events, so they must make progress side by side.  The library probes
 * that once per caller stream (a wait kernel on the caller's stream, then a signal kernel on the side stream) and runs in
 * line by itself -- with one warning on stderr -- when they do not: both on one hardware queue (GPU_MAX_HW_QUEUES=1, many
 * live streams of one priority), serialised dispatch (AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING), a counter-collecting
 * profiler.  A wait that still times out (~10 s) poisons its frame (outputs NaN, never a wrong image), the call renders
 * the frame again in line and the process stays in line from then on; LAE_ELAUNCH only if that fails too.  The reference's
 * loop has no such failure mode (nerf/renderer.py:335-387).  LAE_FRAME_OVERLAP=0 / 1 pins the mode (1 skips the probe). */
LAE_API int lae_render_frame_set_overlap(int on);
/* how the most recent frame ran: 1 = lookahead on the side stream, 0 = in line (switched off, probed as not concurrent, or
 * degraded after a time-out); before the first frame: the configured mode */
LAE_API int lae_render_frame_mode(void);
/* Round 6.  A cross-stream wait that times out while the host is still inside lae_render_frame makes the call render
 * the frame again in line.  One that gives up AFTER the call has returned (only possible without stats_out, which waits for the
 * loop's end) leaves that frame's outputs NaN -- never a wrong image -- and the call has already returned LAE_OK.  After
 * synchronising the frame's stream: 0 = the most recent frame completed, 1 = it was poisoned by such a time-out (render it again;
 * the process runs in line from the next frame on), -1 = no frame yet. */
LAE_API int lae_render_frame_last_status(void);
/* Which side stream: with the first frame of a caller stream the library times 16 hand-overs caller -> side -> caller through the
 * loop's own store / poll kernels on its highest-priority side stream (or the class LAE_FRAME_SIDE_PRIO names) and, when that
 * stream does not run beside the caller's or is slow (~50 instead of ~11 us per hand-over: which hardware queue picks a dispatch
 * up promptly depends on the streams the process already uses), on up to four streams of the caller's class created
 * on demand; the first prompt one is used, else the fastest seen.  us[0..n): the times of the most recent probe in microseconds
 * (candidate 0 first; < 0: not probed / not concurrent); returns the candidate in use, -1 before the first frame. */
LAE_API int lae_render_frame_probe_us(float* us, uint32_t n);
LAE_API int lae_render_frame(const float* rays_o, const float* rays_d, uint32_t N, const float* aabb, float min_near,
                     const uint8_t* grid, const uint8_t* edit_grid, float bound, float dt_gamma, uint32_t max_steps,
                     uint32_t C, uint32_t H, const void* table_f16, const int32_t* offsets, const int32_t* offsets_host,
                     uint32_t L, float S,
                     uint32_t base_resolution, uint32_t gridtype, int align_corners, uint32_t interp,
                     const void* sigma_weights, const void* color_weights, float density_scale, float T_thresh,
                     uint32_t max_n_step, uint64_t row_budget, const float* noises, const float* bg_rays, float bg_r, float bg_g,
                     float bg_b, int blend_bg, int scale_depth, float* weights_sum, float* depth, float* image,
                     float* weights_edit, float* depth_edit, void* workspace, uint64_t workspace_bytes, uint32_t* stats_out,
                     void* stream);

/* MI355X-native extension (no reference counterpart; replaces the host-side
 * `rays_alive = rays_alive[rays_alive >= 0]` + size sync at renderer.py:375,459):
 * order-preserving device-side compaction of the alive list.
 * out_alive[0..n_out) = entries of rays_alive that are >= 0, in order;
 * *n_out_dev (device int32) receives n_out.  scratch: lae_compact_scratch_bytes(n_alive). */
LAE_API uint64_t lae_compact_scratch_bytes(uint32_t n_alive);
LAE_API int lae_compact_rays_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out_alive,
                           int32_t* n_out_dev, void* scratch, void* stream);

/* ------------------------------------------------------------------ */
/* _gridencoder  (gridencoder/src/gridencoder.h:12-15)                */
/* ------------------------------------------------------------------ */

/* gridencoder.cu:448-471  grid_encode_forward(inputs[B,D] f32, embeddings[sO,C],
 * offsets[L+1] i32, outputs[L,B,C], B, D, C, L, S, H, dy_dx[B,L*D*C] or NULL,
 * gridtype, align_corners, interp).  `dtype` selects the element type of
 * embeddings / outputs / dy_dx (the reference dispatches on
 * embeddings.scalar_type(), gridencoder.cu:467). */
LAE_API int lae_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                            void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            float S, uint32_t H, void* dy_dx, uint32_t gridtype,
                            int align_corners, uint32_t interp, int dtype, void* stream);

/* gridencoder.cu:473-503  grid_encode_backward(grad[L,B,C], inputs, embeddings, offsets,
 * grad_embeddings[sO,C] (pre-zeroed, accumulated into), B, D, C, L, S, H,
 * dy_dx or NULL, grad_inputs[B,D] or NULL, gridtype, align_corners, interp) */
LAE_API int lae_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                             const int32_t* offsets, void* grad_embeddings, uint32_t B,
                             uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                             int align_corners, uint32_t interp, int dtype, void* stream);

/* MI355X-native general forms.
 *   blc = 0: [L,B,C] outputs / gradients (the reference's backend layout, and what the fused field op keeps between
 *            encoder and MLP so that no transpose is needed); blc = 1: [B, L*C] directly (what grid.py:57 / grid.py:75
 *            produce with a permute + reshape copy).  Same arithmetic.
 *   in_shift / in_scale: every coordinate is read as (x + in_shift) * in_scale -- GridEncoder.forward's
 *            `(inputs + bound) / (2 * bound)` (grid.py:149) with in_shift = bound, in_scale = fp32(1 / (2 * bound)); pass
 *            0, 1 for coordinates already in [0, 1].
 *   offsets_host: the caller's HOST copy of `offsets` (L + 1 ints) or NULL.  The reference hands the backend a device
 *            tensor only; the module that built it (grid.py:118-127) has the numbers on the host, and passing them lets
 *            the library balance the forward's level -> XCD schedule and skip the launch of the generic atomic kernel
 *            when no level needs it.  It never changes a result: with NULL the forward uses the fixed level l -> XCD
 *            l mod 8 map and the backward always launches the (then idle) companion kernel.  Nothing is cached between
 *            calls (round 1 cached a device->host copy keyed on the device pointer, which a reused address could make
 *            stale). */
LAE_API int lae_grid_encode_forward_ex(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                               uint32_t gridtype, int align_corners, uint32_t interp, int dtype, int blc, float in_shift,
                               float in_scale, const int32_t* offsets_host, void* stream);
LAE_API int lae_grid_encode_backward_ex(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                                int align_corners, uint32_t interp, int dtype, int blc, float in_shift, float in_scale,
                                const int32_t* offsets_host, int32_t* nonfinite_flag, uint32_t* touched_lines, void* stream);
/* nonfinite_flag (may be NULL): device word that is OR-ed with 1 when this call STORES a non-finite value into
 * grad_embeddings (an overflowed sum, a non-finite contribution, or a non-finite value already there that it adds to).  A
 * caller whose gradient buffer is written by these calls only can hand the optimizer's found_inf word (state word 2 of
 * lae_adam_*) and leave the table out of lae_adam_check.  Needs the binned pipeline on every level: D = 3, C = 2,
 * offsets_host given, every level <= 2^21 entries, B <= 2^24 (LAE_EINVAL otherwise).
 * touched_lines (may be NULL; fp16 only, same conditions): bitmap with one bit per 8 table entries (bit k of word w = entries
 * 8 * (32 w + k) ... + 7, i.e. one 64-byte line of the fp32 table); the counting half of the call sets the bits of the lines
 * that hold a corner of a cell some sample lies in (a superset of the lines that receive a non-zero gradient).
 * lae_adam_apply_multi skips lines whose bit was never set (their gradient and both Adam moments are exactly zero).  With
 * the two-call form the bitmap goes to lae_grid_encode_backward_plan (the counting half), not to _planned. */

/* MI355X-native: the binned backward (D = 3, C = 2) in two halves.  Its first half -- the bookkeeping of a counting
 * sort: items per (level, 1024 samples, table partition), their scans -- depends on the sample positions only,
 * so a caller can run it as soon as the positions exist (e.g. right after the march, on another stream beside the
 * forward pass) and hand the result to the second half, which needs the gradients.  plan:
 * lae_grid_backward_plan_bytes(B, L) bytes of device memory owned by the caller, read-only for `_planned` except for its
 * work-queue words, which every execution resets itself (a plan serves any number of executions, one at a time).
 * grad is level-major [L, B, 2] (the layout of lae_grid_encode_backward).  B <= 2^24 samples and L <= 32 (LAE_EINVAL beyond:
 * the kernels address the batch with 32-bit byte offsets); lae_grid_encode_backward itself takes any B and falls back to its
 * global-atomic kernel above that size. */
LAE_API uint64_t lae_grid_backward_plan_bytes(uint32_t B, uint32_t L);
/* 32-bit words of a touched_lines bitmap for a table of n_entries (= offsets[L]) entries: the line bits (+ slack) followed by
 * one "every line of level l is marked" word per level, which the counting pass sets and then stops marking that level.
 * The caller zero-fills the buffer once; filling it with ones means "skip nothing". */
LAE_API uint64_t lae_grid_touched_lines_words(uint64_t n_entries);
LAE_API int lae_grid_encode_backward_plan(const float* inputs, const int32_t* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                  float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                  float in_shift, float in_scale, const int32_t* offsets_host, void* plan,
                                  uint32_t* touched_lines, void* stream);
LAE_API int lae_grid_encode_backward_planned(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings,
                                     uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                     int align_corners, uint32_t interp, int dtype, float in_shift, float in_scale,
                                     const int32_t* offsets_host, const void* plan, int32_t* nonfinite_flag, void* stream);

/* Bytes of LIBRARY workspace the binned backward (D = 3, C = 2) takes for B samples and L levels when it runs both halves
 * itself: the plan + the item queue, sized for the worst case of 8 items per (sample, level) at 10 bytes each for fp16
 * gradients (hashed levels produce 4; only the front of the queue is touched) + the partial sums of split partitions.
 * Informational: the workspace is library-owned (see the header comment: warm up eagerly before capturing). */
LAE_API uint64_t lae_grid_backward_workspace_bytes(uint32_t B, uint32_t L, int dtype);

/* Host-only helper (no GPU call; used by the CPU tests): the level -> XCD schedule of the specialised forward for a table
 * with the given level offsets (HOST pointer, L + 1 ints) and `n_chunks` 256-sample chunks.  nseg_out[8]; segs_out[8][12][3] =
 * (level, first chunk, chunk count) per XCD in execution order.  Returns the largest number of workgroups any XCD owns
 * (the launch is 8 x that), or a negative error code. */
LAE_API int lae_grid_forward_schedule(const int32_t* offsets_host, uint32_t L, float S, uint32_t H, uint32_t n_chunks,
                              uint32_t* nseg_out, uint32_t* segs_out);

/* MI355X-native, A/B switch of the forward for the hot configuration (fp16 table, D = 3, C = 2, linear, hash type):
 * 0 (default) = specialised kernel with the cost-balanced level -> XCD schedule, 1 = specialised kernel with level l on
 * XCD l mod 8, 2 = the generic kernel.  Results are bit-identical in every mode. */
LAE_API int lae_grid_set_forward_mode(int mode);

/* MI355X-native: 0 (default) = binned / LDS-accumulated backward for D = 3, C = 2 (contributions routed to the owner
 * of a 4096-entry table partition, fp16 sums exact, no scattered global atomics), 1 = always the generic kernel (one
 * global atomic per corner, what the reference does). */
LAE_API int lae_grid_set_backward_mode(int mode);

/* gridencoder.cu:639-645  grad_total_variation(inputs[B,D], embeddings, grad, offsets,
 * weight, B, D, C, L, S, H, gridtype, align_corners); inputs has the table dtype. */
LAE_API int lae_grad_total_variation(const void* inputs, const void* embeddings, void* grad,
                             const int32_t* offsets, float weight, uint32_t B, uint32_t D,
                             uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                             int align_corners, int dtype, void* stream);

/* ------------------------------------------------------------------ */
/* _shencoder  (shencoder/src/shencoder.h:9-10)                       */
/* ------------------------------------------------------------------ */

/* shencoder.cu:400-417  sh_encode_forward(inputs[B,3], outputs[B,C*C], B, D=3, C=degree, dy_dx[B,3*C*C] or NULL) */
LAE_API int lae_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D,
                          uint32_t C, float* dy_dx, void* stream);

/* shencoder.cu:419-439  sh_encode_backward(grad[B,C*C], inputs, B, D, C, dy_dx, grad_inputs[B,3] accumulated into) */
LAE_API int lae_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D,
                           uint32_t C, const float* dy_dx, float* grad_inputs, void* stream);

/* ------------------------------------------------------------------ */
/* _ffmlp  (ffmlp/src/ffmlp.h:8-14).  All tensors fp16 (uint16 storage) */
/* ------------------------------------------------------------------ */

/* ffmlp.cu:635-671  ffmlp_forward(inputs[B,in], weights, B, input_dim, output_dim(=16 padded),
 * hidden_dim, num_layers, activation, output_activation, forward_buffer[num_layers,B,hidden], outputs[B,16]).
 * forward_buffer may be NULL.  Round 6: for the shapes the recompute backward serves (hidden 64, ReLU, linear output, 1-2 hidden
 * GEMMs, input 32 / 48 / 64) in the default mode (lae_ffmlp_set_mode(0)) the buffer is NOT written: lae_ffmlp_backward never reads
 * it there and the reference's Python only hands it on (ffmlp.py:31-35).  lae_ffmlp_set_mode(1) fills it like the reference. */
LAE_API int lae_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim,
                      uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                      uint32_t activation, uint32_t output_activation, void* forward_buffer,
                      void* outputs, void* stream);

/* ffmlp.cu:673-709  (inference_buffer[B,hidden] is scratch the reference never reads back) */
LAE_API int lae_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim,
                        uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                        uint32_t activation, uint32_t output_activation, void* inference_buffer,
                        void* outputs, void* stream);

/* ffmlp.cu:749-895  ffmlp_backward(grad[B,16], inputs, weights, forward_buffer, B, ...,
 * calc_grad_inputs, backward_buffer[num_layers,B,hidden], grad_inputs[B,in] or dummy, grad_weights) */
LAE_API int lae_ffmlp_backward(const void* grad, const void* inputs, const void* weights,
                       const void* forward_buffer, uint32_t B, uint32_t input_dim,
                       uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                       uint32_t activation, uint32_t output_activation, int calc_grad_inputs,
                       void* backward_buffer, void* grad_inputs, void* grad_weights,
                       void* stream);
/* MI355X extension (round 5): accumulate != 0 ADDS the weight gradient to grad_weights (an optimizer-owned fp16 accumulator) instead
 * of overwriting it; nonfinite_flag (device int32, may be NULL) is OR-ed with 1 when a stored weight gradient is not finite.  Both
 * are served by the fused recompute backward only (hidden 64, ReLU, 1-2 hidden GEMMs, input 32 / 48 / 64): LAE_EINVAL otherwise. */
LAE_API int lae_ffmlp_backward_ex(const void* grad, const void* inputs, const void* weights,
                          const void* forward_buffer, uint32_t B, uint32_t input_dim,
                          uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                          uint32_t activation, uint32_t output_activation, int calc_grad_inputs,
                          void* backward_buffer, void* grad_inputs, void* grad_weights,
                          int accumulate, int32_t* nonfinite_flag, void* stream);

/* MI355X-native fusion of NeRFNetwork.forward after the encoder (nerf/network_ff.py:57-79): sigma FFMLP(32,64,2 layers)
 * -> sigma = density_scale * exp(h[0]); colour input = [SH degree 4 of dirs | h[1..15] | 0] -> colour FFMLP(32,64,3 layers)
 * -> rgb = sigmoid(out[0..2]).  enc [M,32] fp16, dirs [M,3] fp32, weights in the FFMLP flat fp16 layout, M % 16 == 0.
 * Outputs: h_out [M,16] fp16 (sigma-net output, saved for the backward), sigmas [M] fp32, rgbs [M,3] fp32.
 * enc_level_major != 0: enc is the encoder's native [16, M, 2] layout (lae_grid_encode_forward) instead of [M,32]. */
LAE_API int lae_nerf_head_forward(const void* enc, const float* dirs, const void* sigma_weights, const void* color_weights,
                          uint32_t M, float density_scale, void* h_out, float* sigmas, float* rgbs, int enc_level_major,
                          void* stream);

/* The density query of NeRFNetwork.density (nerf/network_ff.py:83-96) after the encoder: sigma FFMLP -> sigma =
 * density_scale * exp(h[0]); h_out [M,16] fp16 (h[1..15] = geo_feat) may be NULL.  Used by update_extra_state. */
LAE_API int lae_nerf_density_forward(const void* enc, const void* sigma_weights, uint32_t M, float density_scale, void* h_out,
                             float* sigmas, int enc_level_major, void* stream);
/* enc_level_major (round 3): as in lae_nerf_head_forward -- the occupancy-grid maintenance queries the density of 2.1 M
 * cells per cascade and needs neither the [M,32] transpose of the features nor h. */

/* Backward of lae_nerf_head_forward: grad_sigmas [M], grad_rgbs [M,3] fp32 (as produced by
 * lae_composite_rays_train_backward) -> grad_enc [M,32] fp16 (may be NULL), grad_*_weights (fp16, flat FFMLP layout).
 * grad_h is an [M,16] fp16 scratch (receives dL/dh).  Sigmoid and trunc_exp (activation.py:14-17) backward are fused.
 * accumulate_weight_grads != 0: grad_*_weights += dW (the fused optimizer's persistent buffers) instead of = dW.
 * enc_level_major != 0: enc AND grad_enc are [16, M, 2] (what lae_grid_encode_backward consumes directly).  nonfinite_flag (may be NULL): device word OR-ed with 1 when a non-finite WEIGHT gradient is stored (see
 * lae_grid_encode_backward_ex). */
LAE_API int lae_nerf_head_backward(const float* grad_sigmas, const float* grad_rgbs, const void* enc, const float* dirs, const void* h,
                           const float* rgbs, const void* sigma_weights, const void* color_weights, uint32_t M,
                           float density_scale, void* grad_h, void* grad_enc, void* grad_sigma_weights,
                           void* grad_color_weights, int accumulate_weight_grads, int enc_level_major,
                           int32_t* nonfinite_flag, const float* loss_partials, uint32_t loss_n_part, uint32_t loss_n_elem,
                           const float* loss_scale, float* loss_out, void* stream);
/* loss_out != NULL (round 3): the launch that reduces the weight-gradient slabs carries one workgroup more, which finishes a
 * loss value deferred by lae_composite_rays_train_step(defer_loss): loss_out[0] = sum(loss_partials[0 .. loss_n_part)) /
 * loss_n_elem * (*loss_scale, or 1 when NULL), loss_out[1] = the same unscaled -- the bits lae_loss_finish would write. */

/* ---- freqencoder (freqencoder/src/freqencoder.h:6-10; bindings.cpp:5-8) ----
 * outputs [B, C], C = D + 2*D*deg: the input, then per frequency f < deg the D sines and D cosines of x * 2^f.
 * backward: grad [B, C], outputs (saved forward result) -> grad_inputs [B, D] (overwritten). fp32 only. */
LAE_API int lae_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                            void* stream);
LAE_API int lae_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                             float* grad_inputs, void* stream);

/* ---- LAENeRF palette recomposition (editing/style_encoder.py:135-158 forward_train) ----
 * w_logits, o_raw: the [M,16] fp16 outputs of the weight net (first P columns) and offset net (first 3 columns);
 * palette [P,3] fp32 (device); active_mask bit k = palette base k takes part (style_encoder.py `active_palets`).
 *   w_hat [M, popcount(mask)] fp32 = softmax over the active columns, o_hat [M,3] fp16 = tanh,
 *   pred [M,3] fp16 = clamp(w_hat @ palette[active].half() + o_hat, 0, 1).
 * backward: g_pred / g_o fp16 [M,3], g_w fp32 [M, n_active] (each may be NULL = zero) -> g_w_logits, g_o_raw [M,16] fp16
 * (zeros in padded / inactive columns), g_palette [P,3] fp32 (deterministic two-stage reduction);
 * scratch: lae_palette_backward_scratch_bytes(M). */
LAE_API int lae_palette_forward(const void* w_logits, const void* o_raw, const float* palette, uint32_t P, uint32_t active_mask,
                        uint32_t M, void* pred, float* w_hat, void* o_hat, void* stream);
LAE_API uint64_t lae_palette_backward_scratch_bytes(uint32_t M);
LAE_API int lae_palette_backward(const void* w_logits, const void* o_raw, const float* palette, uint32_t P, uint32_t active_mask,
                         uint32_t M, const void* g_pred, const float* g_w, const void* g_o, void* g_w_logits, void* g_o_raw,
                         float* g_palette, void* scratch, void* stream);

/* The point-wise losses of train_LAENeRF_step (nerf/utils.py:990-996; style_encoder.py:183-205) fused behind the
 * recomposition: loss = MSE(pred, target) + w_uniform * max_j sum_i w_hat[i,j] + w_non_uniform * sum_i (1 - max_j w_hat[i,j])
 * + c_offset * sum o_hat^2 [+ the palette-only regulariser `palet_loss` (style_encoder.py:195-202) over all reg_P bases:
 * w_valid * sum floor(p) p + w_distinct * mean_ij (1 - |p_i - p_j|^2 / max), when reg_palette != NULL / (flags & LAE_STYLE_WITH_REG)].
 * forward: fin[12] (device) = {loss * scale, loss, mse, uniform, non_uniform, offset terms, arg-max column, scale, regulariser,
 * ...}; scale: device scalar or NULL (1).  backward: gradients of (upstream * fin[0]) with respect to the two MLP outputs and the
 * palette in ONE kernel (no per-point gradient tensors); upstream: device scalar.  flags: LAE_STYLE_WITH_REG (value 1: what the
 * `with_reg` argument of abi4 meant) | LAE_STYLE_ACCUMULATE_PALETTE (g_palette += instead of =: the caller hands its persistent
 * fp32 gradient buffer over and saves autograd's add launch).
 * scratch: max(lae_style_loss_scratch_bytes(M), lae_palette_backward_scratch_bytes(M)). */
#define LAE_STYLE_WITH_REG 1
#define LAE_STYLE_ACCUMULATE_PALETTE 2
LAE_API uint64_t lae_style_loss_scratch_bytes(uint32_t M);
LAE_API int lae_style_loss_forward(const void* pred, const float* target, const float* w_hat, const void* o_hat, uint32_t M,
                           uint32_t n_active, float w_uniform, float w_non_uniform, float c_offset, const float* scale, float* fin,
                           void* scratch, const float* reg_palette, uint32_t reg_P, float w_valid, float w_distinct, void* stream);
LAE_API int lae_style_loss_backward(const void* w_logits, const void* o_raw, const float* palette, uint32_t P, uint32_t active_mask,
                            uint32_t M, const float* target, const float* fin, const float* upstream, float w_uniform,
                            float w_non_uniform, float c_offset, void* g_w_logits, void* g_o_raw, float* g_palette, void* scratch,
                            int flags, float w_valid, float w_distinct, void* stream);

/* ---- LAENeRF input assembly (editing/style_encoder.py:135-146: encoder rows, SH(3) of the directions, cast, pad, cat -- torch ops
 * and separate launches in the reference) ----
 * forward: feats_lm [16, M, 2] fp16 (the grid kernels' level-major output) + dirs [M,3] fp32 -> feat [Mp, 32] fp16 rows and
 *   off_in [Mp, off_cols] fp16 rows = [feat | SH(degree)(dirs) | zeros]; rows M..Mp-1 zero (Mp >= M, the MLPs' multiple of 16).
 *   degree 0: no directions, off_in is not written.  degree <= 4, off_cols even, 32 + degree^2 <= off_cols <= 48.
 * backward: grad_lm [16, M, 2] fp16 = g_feat [Mp,32] + g_off [Mp,off_cols][:, :32] (fp32 add, one rounding; either may be NULL). */
LAE_API int lae_style_assemble_forward(const void* feats_lm, const float* dirs, uint32_t M, uint32_t Mp, uint32_t degree, void* feat,
                               void* off_in, uint32_t off_cols, void* stream);
LAE_API int lae_style_assemble_backward(const void* g_feat, const void* g_off, uint32_t M, uint32_t off_cols, void* grad_lm, void* stream);

/* ---- edit-grid region growing (editing/editgrid.py:274-340 EditGrid.grow_region_queue; Python + collections.deque
 * in the reference) ----
 * grid: the selection bitfield [C*H^3/8] (Morton order, like density_bitfield), modified in place; density_grid [C, H^3];
 * queue: device array of `capacity` entries x | y << 8 | z << 16 | level << 24; state (device uint32[4]): head, tail
 * (indices into queue; the caller seeds entries [head, tail)), on return also [2] = cells popped, [3] = 1 if the queue
 * would have overflowed (growth stopped early).  Pops at most grow_iterations cells in batches of max_batch (32 in the
 * reference) with the reference's FIFO order, acceptance rule (density >= thresh and not yet selected), neighbour order
 * and level rule; byte writes follow the CPU-tensor semantics of the reference's indexed assignment (last one wins). */
LAE_API int lae_grow_region(uint8_t* grid, const float* density_grid, uint32_t C, uint32_t H, float density_thresh,
                    uint32_t* queue, uint32_t capacity, uint32_t* state, uint32_t grow_iterations, uint32_t max_batch,
                    void* stream);

/* ---- EditDataset transition weights (editing/edit_dataset.py:122-146: torch.cdist in 1000-row chunks + min + clamp_max in the
 * reference) ----
 * out[i] = min(max_dist, min_j |pts[i] - set[j]|) for pts [n,3], set [m,3] fp32 (m == 0: out = max_dist), *out_max = the largest
 * out[i]; scratch: 4 n bytes of device memory.  Squared distances as dx*dx + dy*dy + dz*dz with the last two products fused. */
LAE_API int lae_min_dist_to_points(const float* pts, uint32_t n, const float* set, uint32_t m, float max_dist, float* out,
                           float* out_max, void* scratch, void* stream);

/* ---- occupancy-grid maintenance (nerf/renderer.py:482-649, Python in the reference) ----
 * positions: point j -> xyz = (2 c / (H-1) - 1) * (bound_c - bound_c/H) + (noise * 2 - 1) * bound_c/H and its Morton index
 *   (renderer.py:580-592).  coords NULL: c = (j / H^2, (j / H) % H, j % H) (full sweep, n <= H^3); else coords [n,3] int32.
 *   noise [n,3] in [0,1) or NULL (no jitter). */
LAE_API int lae_density_grid_positions(const int32_t* coords, uint32_t n, uint32_t H, float bound_c, const float* noise,
                               float* xyzs, int32_t* indices, void* stream);
/* partial sweep of update_extra_state on the device (renderer.py:600-612: n random cells + n draws from the occupied cells of
 * density_grid[cas] > 0, found there with nonzero() and a host-sized randint).  Writes 2n points: xyzs [2n,3], indices [2n] (Morton;
 * -1 for the occupied half when no cell is occupied: lae_density_grid_update skips them, the reference keeps the n random points
 * only).  noise [2n,3] in [0,1) or NULL.  Two ways to supply the draws:
 *   rnd == NULL: coords_rand [n,3] int32 in [0,H) and u [n] uniform in [0,1) -- draw j takes occupied cell number
 *     min(floor(u[j] * K), K - 1) in index order, K = their count (tests pin this against the reference's statements);
 *   rnd != NULL: [2, n+1] uniforms in [0,1); both halves are then drawn as SORTED i.i.d. samples (order statistics from partial sums
 *     of exponentials, no sort), cells through Morton codes (H must be a power of two) -- same distribution, points in Morton
 *     order (the encoder and the scatter run 1.7x / 2.5x faster on ordered points).  coords_rand / u are ignored.
 * scratch: lae_density_grid_partial_scratch_bytes(cells, n) bytes of device memory; cells = H^3.  No host read: K stays on the device. */
LAE_API uint64_t lae_density_grid_partial_scratch_bytes(uint32_t cells, uint32_t n);
LAE_API int lae_density_grid_partial_positions(const float* grid_c, uint32_t cells, const int32_t* coords_rand, const float* u, const float* rnd,
                                       uint32_t n, uint32_t H, float bound_c, const float* noise, float* xyzs, int32_t* indices,
                                       void* scratch, void* stream);
/* update: tmp[indices] = sigmas * density_scale (maximum where indices repeat), then on sampled cells with grid >= 0:
 *   grid = max(grid * decay, tmp)  (renderer.py:596, 627, 633-634).  grid [cells] fp32 is one cascade; tmp [cells] uint32
 *   scratch must be zero on entry and is zero again on return. */
LAE_API int lae_density_grid_update(const float* sigmas, const int32_t* indices, uint32_t n, float density_scale, float decay,
                            uint32_t cells, float* grid, uint32_t* tmp, void* stream);
/* mark_untrained_grid (renderer.py:482-554): grid[cas, morton(c)] = -1 for cells seen by no camera or closer than
 *   min_near to one.  poses [B,4,4] fp32 camera-to-world, intrinsics (fx, fy, cx, cy), grid [C, H^3] fp32 in place. */
LAE_API int lae_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t C, uint32_t H,
                            float bound, float min_near, int filter_close_point, float* grid, void* stream);

/* ---- fused Adam + GradScaler (torch.optim.Adam / torch.cuda.amp.GradScaler in the reference: main_nerf.py:223,
 * nerf/utils.py:1474-1482).  `state` is a 64-byte device block:
 *   [0] f32 scale  [1] i32 growth_tracker  [2] i32 found_inf  [3] i32 skip  [4] i32 step  [5] f32 1/(1-beta1^step)
 *   [6] f32 sqrt(1-beta2^step)  [7] f32 1/scale of this step  [8] i32 skipped steps
 *   [9] [10] f32 the bias corrections of step [11] (i32; 0 = none) computed with the betas [12] [13] -- left by the apply launch
 *   so that the one-thread begin launch on the step's critical path does no double-precision pow (round 3; begin computes
 *   them itself when [11] is not the step it is about to take)  [14..15] reserved.  The apply entry points therefore WRITE
 *   words 9-13 of `state` although they take it as const.
 * Per step: lae_adam_check on every gradient, ONE lae_adam_begin, lae_adam_apply on every parameter.
 * check: found_inf |= any non-finite element (grad fp16 or fp32, 16-byte aligned). */
LAE_API int lae_adam_check(const void* grad, int grad_is_half, uint64_t n, void* state, void* stream);
/* begin: found_inf -> skip (scale *= backoff, tracker = 0) or step += 1 with bias corrections (double precision) and
 * scale growth after `growth_interval` finite steps (grad_scaler.py _amp_update_scale_).  use_scaler = 0: plain Adam. */
LAE_API int lae_adam_begin(void* state, float beta1, float beta2, int growth_interval, float growth_factor, float backoff_factor,
                   int use_scaler, void* stream);
/* apply: g = grad / scale; Adam update of param / exp_avg / exp_avg_sq (fp32, torch.optim.Adam formulas); optional fp16
 * copy of the new parameters into shadow_half (the table the grid encoder gathers from); grad is zeroed.  lr is read
 * from device memory.  On a skipped step only the gradient is zeroed. */
LAE_API int lae_adam_apply(float* param, float* exp_avg, float* exp_avg_sq, void* grad, int grad_is_half, void* shadow_half, uint64_t n,
                   const void* state, const float* lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* Trainer criterion + loss scaling in one kernel (nerf/utils.py train_step: MSELoss(reduction='none')(pred, gt).mean(-1)
 * .mean(), then GradScaler.scale): loss_out[0] = mean((pred - target)^2) * scale, loss_out[1] = the unscaled loss,
 * grad[i] = d loss_out[0] / d pred[i].  scale: device float (FusedAdam state word 0) or NULL (= 1).  fp32, n elements. */
LAE_API int lae_mse_loss_forward(const float* pred, const float* target, uint32_t n, const float* scale, float* loss_out, float* grad,
                         void* stream);

/* multi-tensor forms of check / apply: host arrays of n_tensors (<= 8) device pointers / sizes; one launch each.
 * touched_lines (may be NULL, entries may be NULL): per tensor the bitmap lae_grid_encode_backward_ex maintains (one bit per 16
 * parameters); with weight_decay == 0 a line whose bit is clear is neither read nor written -- Adam's update of parameters whose
 * gradient, exp_avg and exp_avg_sq are all zero is exactly zero. */
LAE_API int lae_adam_check_multi(uint32_t n_tensors, const void* const* grads, const int* grad_is_half, const uint64_t* sizes, void* state,
                         void* stream);
LAE_API int lae_adam_apply_multi(uint32_t n_tensors, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs, void* const* grads,
                         const int* grad_is_half, void* const* shadows_half, const uint64_t* sizes, const float* const* lrs, const void* const* touched_lines,
                         const void* state, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* torch_ema.ExponentialMovingAverage.update() of the reference's trainer (nerf/utils.py:407-408 construct, :1502-1503 update once per
 * epoch; decay 0.95, main_nerf.py:244): shadow -= one_minus_decay * (shadow - param) for up to 8 fp32 tensors in one launch.  The
 * caller computes one_minus_decay = 1 - min(decay, (1 + num_updates) / (10 + num_updates)) like torch_ema does. */
LAE_API int lae_ema_update_multi(uint32_t n_tensors, float* const* shadows, const float* const* params, const uint64_t* sizes,
                         float one_minus_decay, void* stream);

/* MI355X-native: 0 (default) = fused backward (activations recomputed in registers, forward_buffer /
 * backward_buffer untouched: both are scratch the reference's Python never reads), every wave accumulating all dW tiles
 * over its own rows, operands transposed on the matrix cores (round 3); 1 = always the three-kernel path that fills both
 * buffers exactly like the reference; 3 = fused backward with the dW tiles divided among the waves of a workgroup (the
 * round-2 kernel, kept as the one A/B predecessor).  Any other value: LAE_EINVAL (round 4 removed the first fused design,
 * mode 2, and the superseded fused-head forward kernels, modes 16-18). */
LAE_API int lae_ffmlp_set_mode(int mode);

/* ffmlp.cu:721-740: the reference keeps process-global side streams for its
 * split-K CUTLASS GEMMs.  The MFMA backward reduces dW inside one stream, so
 * these are accepted and ignored (kept so FFMLP.__init__ runs unmodified). */
LAE_API int lae_allocate_splitk(uint64_t size);
LAE_API int lae_free_splitk(void);

/* MI355X-native: library workspaces (see the header comment).  lae_free_workspaces() frees the live and the retired
 * buffers of every device after synchronising it; the caller guarantees that no captured graph that used the library
 * is replayed afterwards.  lae_workspace_bytes(0 / 1) = bytes currently held live / retired. */
LAE_API int lae_free_workspaces(void);
LAE_API uint64_t lae_workspace_bytes(int retired);

#ifdef __cplusplus
}
#endif
#endif /* LAENERF_H */
