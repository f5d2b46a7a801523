#!/usr/bin/env python3
"""bench.py -- lego 800x800 train-step throughput of the MI355X hot path (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of 4096 synthetic rays:
  near_far_from_aabb -> march_rays_train -> hash-grid encode (L=16, T=2^19, fp16 table under autocast) ->
  sigma FFMLP (2x64) -> trunc_exp -> SH(4) -> colour FFMLP (3x64) -> sigmoid -> composite_rays_train ->
  MSE -> backward of all of it (composite, MLPs, grid scatter) -> Adam step + GradScaler
i.e. the reference's Trainer.train_step/backward/optimizer.step sequence (nerf/utils.py:1474-1482) in steady
state (mean_count > 0, no host sync inside the step).  Synthetic rays / occupancy / weights:
no dataset or checkpoint exists offline.  value = all ranks' rays / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from laenerf_amd.streams import PROBES as SIDE_STREAM_PROBES, concurrent_side_stream      # noqa: E402  (no GPU call at import; why: its docstring)

GRID_FWD_BYTES_FP16 = 588        # SURVEY.md 8d / BASELINE.md 4: 12 + 16*8*2*2 + 16*2*2 bytes per sample
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS_F16 = 2500.0    # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (the 2:1-sparsity figure is never used)
# fused head: sigma net 32->64->64->16 + colour net 32->64->64->64->16, 2 FLOP per multiply-add
HEAD_FWD_FLOP = 2 * (32 * 64 + 64 * 64 + 64 * 16) + 2 * (32 * 64 + 64 * 64 + 64 * 64 + 64 * 16)      # 36 864 per sample
HEAD_BWD_FLOP = 2 * HEAD_FWD_FLOP                                                                    # dX and dW products: 73 728


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=40)
    p.add_argument("--rays", type=int, default=4096)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-frame", action="store_true", help="skip the 800x800 inference-frame timing (extra field eval_frame)")
    p.add_argument("--no-style", action="store_true", help="skip the LAENeRF palette-network step timing (extra field style_step)")
    p.add_argument("--no-dropin", action="store_true", help="skip the zero-edit drop-in train step (extra field drop_in_step)")
    p.add_argument("--cpu-rays", type=int, default=0, help="rays in the CPU-baseline sample (0 = auto, ~15 s)")
    p.add_argument("--no-optimizer", action="store_true", help="diagnostic only: skip Adam/GradScaler (not the reported metric)")
    p.add_argument("--workload", choices=["train", "frame1080", "flower"], default="train",
                   help="train (default): the BASELINE configs[1] train step.  frame1080: configs[3]-style whole-frame render, "
                        "1920x1080 rays sharded over the ranks in 128-ray tiles, one all-gather (RCCL) per frame; a step is a frame.  "
                        "flower: only the configs[2]-shaped train step of the default line's `flower_step` (for profiling)")
    p.add_argument("--dp", action="store_true",
                   help="data-parallel training (not the default metric mode): every rank trains on its own ray batch, gradients "
                        "are averaged with ONE flat all-reduce (RCCL) per dtype before the Adam step (laenerf_amd/dist.py)")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying the captured HIP graph")
    p.add_argument("--steps-per-graph", type=int, default=0,
                   help="pipelined mode: consecutive train steps captured into one graph replay (at most 16; reduced to a divisor "
                        "of --steps; 0 = the largest divisor of --steps that is <= 16).  A graph boundary costs 15-18 us of device "
                        "time on this stack")
    p.add_argument("--march-beside", choices=["forward", "backward"], default="backward",
                   help="pipelined mode: which half of step k the march of step k+1 runs beside")
    p.add_argument("--no-pipeline", action="store_true",
                   help="one graph per step; default: the march of step k+1 (no weight dependence) is its own graph, replayed on a "
                        "side stream beside the shading / backward / optimizer graph of step k")
    p.add_argument("--torch-loss", action="store_true", help="A/B: torch mse_loss + scale() instead of laenerf_amd.losses.mse_loss_scaled")
    p.add_argument("--torch-optimizer", action="store_true",
                   help="A/B: torch.optim.Adam(fused) + torch.amp.GradScaler instead of laenerf_amd.optim.FusedAdam")
    return p.parse_args()


def host_threads():
    """threads the CPU baselines run on: the cores this job may actually use -- the cgroup CPU quota when there is one (the GPU
    pool gives a one-GPU job 16 CPUs' worth of time on a 256-thread host: /sys/fs/cgroup/cpu.max = "1600000 100000"; 256 threads
    under that quota are throttled, measured 13.96 s per cfg1 step against 1.11 s on 16 threads), else the affinity mask / core
    count.  LAE_CPU_THREADS overrides."""
    if os.environ.get("LAE_CPU_THREADS"):
        return max(1, int(os.environ["LAE_CPU_THREADS"]))
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(n_rays_hint, n_threads):
    """The oracle (C restatement of the reference kernels, oracle/lae_oracle.c) timed on the host cores on a
    bounded sample of the SAME workload: forward + backward of one train step, without the optimizer."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from laenerf_amd import synthetic as S
    O.build()
    bits = S.pack_bits_np(S.sphere_density_grid(), 10.0)
    offsets, pls = O.grid_offsets(num_levels=16, desired_resolution=2048)
    rng = np.random.default_rng(0)
    table_h = O.to_f16_bits(rng.uniform(-1e-4, 1e-4, (int(offsets[-1]), 2)).astype(np.float32))
    ws = O.to_f16_bits(rng.uniform(-0.2165, 0.2165, O.ffmlp_num_params(32, 64, 2)).astype(np.float32))
    wc = O.to_f16_bits(rng.uniform(-0.2165, 0.2165, O.ffmlp_num_params(32, 64, 3)).astype(np.float32))

    def one_chunk(o, d, noises):
        n = o.shape[0]
        nears, fars = O.near_far_from_aabb(o, d, [-1, -1, -1, 1, 1, 1], 0.2)
        xyzs, dirs, deltas, rays, counter = O.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, noises, M=n * 1024)
        m = int(counter[0]); m += 128 - m % 128
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        enc, _ = O.grid_encode_forward((xyzs + 1) / 2, table_h, offsets, pls, 16, f16=True, out_blc=True)
        h, fb1 = O.ffmlp_forward(enc, ws, 32, 16, 64, 2)
        hf = O.from_f16_bits(h)
        sigma = np.exp(hf[:, 0])
        sh, _ = O.sh_encode_forward(dirs, 4)
        cin = O.to_f16_bits(np.concatenate([sh, hf[:, 1:], np.zeros((m, 1), np.float32)], 1))
        oc, fb2 = O.ffmlp_forward(cin, wc, 32, 16, 64, 3)
        rgb = 1 / (1 + np.exp(-O.from_f16_bits(oc)[:, :3]))
        wsum, depth, image = O.composite_rays_train_forward(sigma, rgb, deltas, rays, 1e-4)
        gimg = (2 * (image + (1 - wsum)[:, None] - 0.5) / image.size).astype(np.float32)
        gws = -gimg.sum(1)
        gs, gc = O.composite_rays_train_backward(gws, gimg, sigma, rgb, deltas, rays, wsum, image, 1e-4)
        goc = np.zeros((m, 16), np.float32); goc[:, :3] = gc * rgb * (1 - rgb)
        _, gcin, _ = O.ffmlp_backward(O.to_f16_bits(goc), cin, wc, fb2, 32, 16, 64, 3, calc_grad_inputs=True)
        gh = np.zeros((m, 16), np.float32); gh[:, 0] = gs * sigma; gh[:, 1:] = O.from_f16_bits(gcin)[:, 16:31]
        _, genc, _ = O.ffmlp_backward(O.to_f16_bits(gh), enc, ws, fb1, 32, 16, 64, 2, calc_grad_inputs=True)
        O.grid_encode_backward(genc, (xyzs + 1) / 2, (int(offsets[-1]), 2), offsets, pls, 16, f16=True, grad_blc=True)
        return int(counter[0])

    o, d = S.lego_like_rays(4096, seed=0, n_views=1)
    noises = np.random.default_rng(1).random(4096).astype(np.float32)
    # calibrate on 64 rays, then size the sample for ~15 s of wall time on n_threads threads: whole 4096-ray steps
    # (different ray batches) when one step is too short for the box
    t0 = time.perf_counter(); one_chunk(o[:64], d[:64], noises[:64]); t_cal = time.perf_counter() - t0
    if n_rays_hint <= 0:
        # threads do not scale linearly (every chunk zero-fills its own 24 MB gradient table): assume 1/4 efficiency
        n_rays = int(max(256, 15.0 / (t_cal / 64) * n_threads / 4))
        n_rays = min(n_rays, 16 * 4096)
    else:
        n_rays = n_rays_hint
    n_steps = max(1, (n_rays + 4095) // 4096)
    work = []
    for k in range(n_steps):
        ok_, dk_ = (o, d) if k == 0 else S.lego_like_rays(4096, seed=k, n_views=1)
        n_k = min(4096, n_rays - 4096 * k)
        for c in np.array_split(np.arange(n_k), max(1, min(n_threads, n_k // 16))):
            work.append((ok_[c], dk_[c], noises[c]))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_threads) as ex:
        tot = sum(ex.map(lambda a: one_chunk(*a), work))
    dt = time.perf_counter() - t0
    # also the ONE-thread figure (same chunks of 16+ rays, sequential, ~3 s) and the CPU's name
    n1 = int(min(n_rays, max(64, 3.0 / (t_cal / 64))))
    t1 = time.perf_counter()
    for c in np.array_split(np.arange(n1), max(1, n1 // 64)):
        one_chunk(o[c], d[c], noises[c])
    dt1 = time.perf_counter() - t1
    return {"value": round(n_rays / dt / 1e6, 6), "unit": "Mrays/s", "cores": n_threads, "kind": "port",
            "cpu_model": cpu_model(), "one_thread": {"value": round(n1 / dt1 / 1e6, 6), "unit": "Mrays/s", "rays": n1, "seconds": round(dt1, 1)},
            "sample": f"{n_rays} rays = {n_steps} step(s) of the same 4096-ray workload ({tot} samples), forward+backward "
                      f"without optimizer, oracle/lae_oracle.c on {n_threads} threads, {dt:.1f} s"}


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def roofline_more(tm, dev, with_frame=True):
    """the kernels furthest below their roofline, priced like `roofline` (HIP events on the launch stream around each operator in
    20 eager steps): hash-grid backward against HBM on its 588 algorithmic bytes per sample (fill + accumulate
    on the main stream, and with the counting pass + scans that run ahead on the side stream), fused head forward / backward
    against the dense fp16 MFMA peak, and the same encoder kernel on the tile-ordered rows of an inference frame."""
    def per(name):
        e = tm.get(name)
        if not e or not e["calls"]:
            return None
        return e["ms"] / e["calls"] * 1e3, e["units"] / e["calls"]            # us per launch, units per launch
    out = {}
    gb, gp = per("grid_encode_backward"), per("grid_backward_plan")
    if gb:
        us_main, n = gb
        us_all = us_main + (gp[0] if gp else 0.0)
        for key, us in (("grid_backward_main_stream", us_main), ("grid_backward_total", us_all)):
            gbs = n * GRID_FWD_BYTES_FP16 / (us * 1e-6) / 1e9
            out[key] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "avg_us": round(us, 2), "samples_per_launch": int(n), "bytes_per_sample": GRID_FWD_BYTES_FP16}
        out["grid_backward_main_stream"]["kernels"] = "k_bwd_walk<FILL> + k_bwd_acc"
        out["grid_backward_total"]["kernels"] = "k_bwd_walk<COUNT> + k_bwd_scan_units/parts (side stream, ahead) + k_bwd_walk<FILL> + k_bwd_acc"
    for key, name, flop, kern in (("head_forward", "nerf_head_forward", HEAD_FWD_FLOP, "k_nerf_head_fwd5"),
                                  ("head_backward", "nerf_head_backward", HEAD_BWD_FLOP, "k_mlp_bwd_wave x2 + k_dw_reduce2")):
        e = per(name)
        if e:
            us, n = e
            tf = n * flop / (us * 1e-6) / 1e12
            out[key] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS_F16, 4),
                        "avg_us": round(us, 2), "samples_per_launch": int(n), "flop_per_sample": flop, "kernels": kern}
    # the encoder on tile-ordered frame rows: the operator loop of one 800x800 frame goes through the timed backend operator
    # (skipped with --no-frame: the kernel traces of the train step stay free of inference kernels)
    if not with_frame:
        out["timed"] = "HIP events on the launch stream around each operator, 20 eager steps after the timed region"
        return out
    try:
        from laenerf_amd import backend, synthetic as S
        net, r = eval_model(dev)
        o, d = S.frame_rays(800, 800)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            r.render_eval(o, d, bg_color=1, max_steps=1024, image_hw=(800, 800), frame_loop=False)       # warm-up
            with backend.kernel_timing(only=("grid_encode_forward",)) as kt:
                r.render_eval(o, d, bg_color=1, max_steps=1024, image_hw=(800, 800), frame_loop=False)
        e = kt.result.get("grid_encode_forward")
        if e and e["calls"]:
            gbs = e["units"] * GRID_FWD_BYTES_FP16 / (e["ms"] * 1e-3) / 1e9
            out["frame_encoder"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                    "launches": e["calls"], "rows": int(e["units"]), "sum_ms": round(e["ms"], 3), "bytes_per_sample": GRID_FWD_BYTES_FP16,
                                    "kernels": "k_grid_fwd_lean on the tile-ordered rows of one 800x800 inference frame (operator loop, summed over its iterations)"}
    except Exception as ex:                                    # a diagnostic must never take the line down
        out["frame_encoder"] = {"error": repr(ex)}
    out["timed"] = "HIP events on the launch stream around each operator, 20 eager steps after the timed region (frame_encoder: one operator-loop frame)"
    return out


def eval_model(dev, bound=1, seed=4321):
    """the model every inference number of this file is quoted on: fixed seed, default initialisation, analytic occupancy
    -- independent of --steps (round 1 rendered with the network as trained by the timed steps, so the frame time moved
    with K)"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(seed)
    net = NeRFNetwork(bound=bound).to(dev).eval()
    r = NeRFRenderer(net, bound=bound, min_near=0.2).to(dev).eval()
    C = r.cascade
    grid = S.sphere_density_grid(cascade=C, bound=float(bound)) if bound == 1 else S.flower_density_grid()
    r.density_bitfield = torch.from_numpy(S.pack_bits_np(grid, 10.0)).to(dev)
    return net, r


def eval_frame(dev, H=800, W=800, density_scale=1.0):
    """whole-frame inference render (the march_rays / composite_rays loop of run_cuda, renderer.py:335-387) of one 800x800
    view with the fixed model of eval_model(); median of 5 frames after one warm-up.  `ms_per_frame`: the loop as
    ONE backend call with its state on the device (lae_render_frame, reference schedule); `operator_loop_ms`: the same
    kernels driven operator by operator from Python like the reference's loop (one host read of n_alive per iteration)."""
    from laenerf_amd import synthetic as S
    net, r = eval_model(dev)
    r.density_scale = density_scale                           # > 1: a trained-shaped scene (rays saturate at a surface; `eval_frame_surface`)
    o, d = S.frame_rays(H, W)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)

    def timed(**kw):
        times, res = [], None
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
                res = r.render_eval(o, d, bg_color=1, max_steps=1024, image_hw=(H, W), **kw)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return sorted(times[1:])[2], res
    t, res = timed(frame_loop=True, want_stats=True)
    t_op, res_op = timed(frame_loop=False)
    return {"ms_per_frame": round(t * 1e3, 2), "rays": H * W, "Mrays_per_s": round(H * W / t / 1e6, 2),
            "Msamples_per_s": round(res["stats"]["rows"] / t / 1e6, 1),
            "rays_hitting_geometry": round(float((res["weights_sum"] > 0).float().mean()), 3),
            "iterations": res["stats"]["iterations"], "samples_through_network": res["stats"]["rows"],
            "operator_loop_ms": round(t_op * 1e3, 2),
            "max_abs_image_diff_vs_operator_loop": float((res["image"] - res_op["image"]).abs().max()),
            "side_stream": _frame_probe(),                       # which side-stream candidate the frame loop chose, and the timed hand-overs
            "density_scale": density_scale,
            "note": "800x800 inference render of the fixed eval model (seed 4321, default init: independent of --steps), T_thresh 1e-4, "
                    "device-resident loop (lookahead marcher on a side stream)" +
                    ("" if density_scale == 1.0 else f"; density scale {density_scale:g} as in edit_extract: rays end at a surface (early-termination "
                                                     "branches of k_frame_head and the compaction carry the frame)")}


def _frame_probe():
    from laenerf_amd.backend import raymarching_backend
    return raymarching_backend.render_frame_probe()


def sharded_frame1080(world, rank, dev, backend_name, steps, warmup, with_n1=False):
    """configs[3]: full-frame inference render, rays sharded across the ranks (laenerf_amd/dist.py), one all-gather of the
    [n/W, 5] fp32 block per frame (north_star / SURVEY 8e; the reference's dormant `dist.all_gather(preds)`, nerf/utils.py:1555-1570).
    Every rank holds the same (random-init, synthetic-occupancy) model; a step = a frame.  Collective: every rank calls it;
    returns the same dict on every rank.  with_n1: rank 0 also renders the whole frame alone (W = 1, the reference's schedule)
    while the others wait, for `speedup_vs_n1`."""
    import hashlib
    import torch.distributed as dist
    from laenerf_amd import synthetic as S
    from laenerf_amd.dist import render_frame_sharded, broadcast_model_state
    # configs[3] is mip360/bonsai: scripts/configs_mip360/bonsai.sh has bound=2 -> 2 cascades (renderer.py:74), a 512 KiB
    # bitfield and a 6 328 848-entry table (finest resolution 4096); the camera orbits INSIDE the bound-2 box
    net, r = eval_model(dev, bound=2, seed=1234)              # identical replicas on every rank
    broadcast_model_state(r, src=0)                           # replicate table / MLPs / bitfield at load
    H, W = 1080, 1920
    o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)

    # rows per iteration of the device-resident loop: the reference's rule sizes an iteration by the rays of THE CALL
    # (n_step = max(min(N // n_alive, 8), 1), renderer.py:363), so a rank holding 1/W of the rays would run the whole frame's
    # iteration count on 1/W of the work per iteration -- 177 launch-chain latencies for a ninth of the samples (`frame1080.
    # shards_of_8.reference_rule` of the N = 1 line).  With W > 1 every rank therefore takes the WHOLE frame's row budget
    # (row_budget = W x its rays): same per-ray sample sequences, <= 1e-5 image difference (rays_t rounding at other boundaries:
    # tests/test_gpu_frame.py::test_frame_loop_row_budget), a few times fewer iterations.  LAE_FRAME_REFERENCE_SCHEDULE=1 keeps
    # the reference's rule on every rank.
    budget = 0 if (world == 1 or os.environ.get("LAE_FRAME_REFERENCE_SCHEDULE") == "1") else None

    def render(ro, rd, whole=None):
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            return r.render_eval(ro, rd, bg_color=1, max_steps=1024,
                                 row_budget=(whole if whole is not None else world * ro.shape[0] if budget is None else budget))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def frame():
        return render_frame_sharded(render, o, d, rank, world, image_hw=(H, W))
    for _ in range(max(warmup, 2)):
        res = frame()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = frame()
    sync()
    dt = time.perf_counter() - t0
    frame_hash = hashlib.sha256(res["image"].cpu().numpy().tobytes()).hexdigest()[:16]
    hashes = [frame_hash]
    n1_ms = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev if backend_name == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        hashes = [None] * world
        dist.all_gather_object(hashes, frame_hash)               # every rank holds the whole frame after the all-gather
        if with_n1:
            if rank == 0:                                        # the same model, the whole frame on ONE GPU, the others idle
                n1_all = {}
                for k in (0, 2, 4, 8):                           # 0 = the reference's rule; k x N rows per iteration otherwise
                    ts = []
                    for it in range(5):
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        render_frame_sharded(lambda ro, rd: render(ro, rd, whole=k * H * W), o, d, 0, 1, image_hw=(H, W))
                        torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
                    n1_all["reference_rule" if k == 0 else f"{k}N"] = sorted(ts[2:])[1] * 1e3
                n1_ms = n1_all["reference_rule"]
            sync()
    ms = dt / steps * 1e3
    tiles = -(-H * W // 128)
    out = {"ms_per_frame": round(ms, 3), "Mrays_per_s": round(H * W * steps / dt / 1e6, 3), "rays": H * W, "frames_timed": steps,
           "n_gpus": world, "frame_sha256_per_rank": hashes, "ranks_hold_the_same_frame": len(set(hashes)) == 1,
           "gather_bytes_per_rank": int(-(-tiles // world) * 128 * 5 * 4),
           "collective": "one all_gather_into_tensor of the [n/W, 5] fp32 block per frame" if world > 1 else "none (W = 1)",
           "backend": (dist.get_backend() if world > 1 else None), "world_size": (dist.get_world_size() if world > 1 else 1),
           "rays_hitting_geometry": round(float((res["weights_sum"] > 0).float().mean()), 3),
           "parallelism": f"rays in 8x8-pixel-tile order, 128-ray units (two pixel tiles) dealt round-robin to {world} rank(s), one all-gather of [n/W,5] fp32 per frame"}
    if n1_ms is not None:
        best = min(n1_all.values())
        out["n1_ms_per_frame"] = round(n1_ms, 3)                 # the reference's schedule (renderer.py:363)
        out["n1_row_budget_ms"] = {k: round(v, 3) for k, v in n1_all.items()}
        out["best_n1_ms_per_frame"] = round(best, 3)             # the fastest one-GPU schedule: what strong scaling is quoted against
        out["speedup_vs_n1"] = round(best / ms, 3)
        out["speedup_vs_n1_reference_schedule"] = round(n1_ms / ms, 3)
    return out


def frame_workload(args, world, rank, dev, backend_name):
    """`--workload frame1080`: the sharded frame as the bench's own `value` (strong scaling: the frame is fixed)"""
    import torch.distributed as dist
    f = sharded_frame1080(world, rank, dev, backend_name, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "frame_sha256_per_rank": f["frame_sha256_per_rank"], "ranks_hold_the_same_frame": f["ranks_hold_the_same_frame"],
            "gather_bytes_per_rank": f["gather_bytes_per_rank"],
            "metric": "Mrays/s, 1920x1080 whole-frame inference render", "value": f["Mrays_per_s"], "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": f["ms_per_frame"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16 (table, MLP) / f32 (march, composite)", "data": "synthetic",
            "config": {"workload": "configs[3]-shaped (mip360/bonsai: bound 2, 2 cascades, 512 KiB bitfield, 6 328 848-entry table): "
                                   "1920x1080 rays of one view from inside the box, L=16 T=2^19 hash grid + 2x64 / 3x64 ffmlp, "
                                   "analytic occupancy, device-resident inference loop per rank",
                       "rays_per_frame": f["rays"], "rays_hitting_geometry": f["rays_hitting_geometry"],
                       "backend": f["backend"], "world_size": f["world_size"], "parallelism": f["parallelism"]}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def frame1080(dev, frames=5):
    """configs[3] on ONE GPU, inside the default line so that the driver times it: the 1920x1080 bonsai-shaped frame of
    `--workload frame1080` (bound 2, 2 cascades, 6 328 848-entry table, camera inside the box) through
    dist.render_frame_sharded at W = 1, median of `frames` frames after two warm-ups; plus EVERY rank's share of an 8-way split
    (rank r: tiles r, r + 8, r + 16, ...: what each of 8 GPUs would render before the 5.18 MB all-gather) timed alone on this GPU,
    one after the other, and the exchange around it (frame_exchange)."""
    from laenerf_amd import synthetic as S
    from laenerf_amd.dist import render_frame_sharded
    net, r = eval_model(dev, bound=2, seed=1234)
    H, W = 1080, 1920
    o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    stats = {}

    budget = {"rows": 0}

    def render(ro, rd):
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            res = r.render_eval(ro, rd, bg_color=1, max_steps=1024, want_stats=True, row_budget=budget["rows"])
        stats.update(res["stats"])
        return res

    def timed(fn):
        ts = []
        for it in range(frames + 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts[2:])[len(ts[2:]) // 2], out
    t, res = timed(lambda: render_frame_sharded(render, o, d, 0, 1, image_hw=(H, W)))
    whole = dict(stats)
    # the SAME whole frame on this one GPU with k x N rows per iteration (the boosted schedule a shard of a W-rank job runs,
    # row_budget = W x its rays): strong scaling is quoted against the BEST one-GPU time, not against the reference rule's
    boosted = {}
    for k in (2, 4, 8):
        budget["rows"] = k * H * W
        tk, resk = timed(lambda: render_frame_sharded(render, o, d, 0, 1, image_hw=(H, W)))
        boosted[f"{k}N"] = {"ms": round(tk * 1e3, 2), "iterations": stats["iterations"],
                            "max_abs_image_diff_vs_reference_schedule": float((resk["image"] - res["image"]).abs().max())}
    budget["rows"] = 0
    best_key = min(boosted, key=lambda k: boosted[k]["ms"])
    best_ms = min(t * 1e3, boosted[best_key]["ms"])
    # ---- all eight shards of an 8-way split, one after the other on this one GPU: what each of 8 GPUs would
    # render before the all-gather; the projection rests on the SLOWEST one plus the measured exchange
    from laenerf_amd.dist import frame_plan
    shards = {"boosted": [], "reference_rule": []}
    for rk in range(8):
        take = frame_plan(H * W, rk, 8, dev, (H, W))["take"]
        ot, dt_ = o[take], d[take]
        for mode, rows in (("reference_rule", 0), ("boosted", H * W)):      # boosted = the whole frame's row budget (what --gpus 8 runs)
            budget["rows"] = rows
            tk, _ = timed(lambda: render(ot, dt_))
            shards[mode].append({"rank": rk, "ms": round(tk * 1e3, 3), "iterations": stats["iterations"], "samples": stats["rows"]})
    budget["rows"] = 0

    def spread(key, mode):
        v = [e[key] for e in shards[mode]]
        return {"max": max(v), "mean": round(float(np.mean(v)), 3), "min": min(v)}
    exch = frame_exchange(dev, H, W)
    slow_b, slow_r = spread("ms", "boosted")["max"], spread("ms", "reference_rule")["max"]
    ex_ms = exch.get("total_ms_per_frame", 0.0) or 0.0
    return {"ms_per_frame": round(t * 1e3, 2), "rays": H * W, "Mrays_per_s": round(H * W / t / 1e6, 2),
            "iterations": whole["iterations"], "samples_through_network": whole["rows"],
            "Msamples_per_s": round(whole["rows"] / t / 1e6, 1),
            "rays_hitting_geometry": round(float((res["weights_sum"] > 0).float().mean()), 3),
            "reference_schedule_ms": round(t * 1e3, 2), "row_budget_ms": boosted,
            "best_n1_ms_per_frame": round(best_ms, 2),
            "best_n1_schedule": ("reference rule (N rows per iteration)" if best_ms == t * 1e3 else f"{best_key} rows per iteration"),
            "shards_of_8": {"rays_per_shard": int(-(-(-(-H * W // 128)) // 8) * 128),
                            "boosted_budget": {"ms": spread("ms", "boosted"), "samples": spread("samples", "boosted"),
                                               "iterations": spread("iterations", "boosted"), "per_rank_ms": [e["ms"] for e in shards["boosted"]]},
                            "reference_rule": {"ms": spread("ms", "reference_rule"), "iterations": spread("iterations", "reference_rule"),
                                               "per_rank_ms": [e["ms"] for e in shards["reference_rule"]]},
                            "note": "every rank's tiles of the 8-way round-robin split of the pixel-tile ray order, rendered alone on this one GPU "
                                    "(median of 5 each), with the whole frame's row budget per iteration (what `--workload frame1080 --gpus 8` runs "
                                    "per rank) and with the reference's per-call rule (renderer.py:363)"},
            "exchange": exch,
            "projected_speedup_at_8": {"vs_best_n1": round(best_ms / (slow_b + ex_ms), 2),
                                       "reference_rule_both_sides": round(t * 1e3 / (slow_r + ex_ms), 2),
                                       "slowest_shard_ms": slow_b, "exchange_ms": round(ex_ms, 3),
                                       "without_exchange_from_rank0_only": round(best_ms / shards["boosted"][0]["ms"], 2),
                                       "note": "one GPU's whole-frame time / (SLOWEST of the eight shards + the exchange measured at W = 1 on RCCL: "
                                               "collective launch, the rank's ray gather, block pack, the gather into the caller's order on the "
                                               "full-size buffer); the 7-link xGMI transfer of 5.18 MB per rank (~34 us at 7 x 153 GB/s, SURVEY 8e) "
                                               "is NOT in it: no multi-GPU node.  `vs_best_n1` divides the fastest one-GPU schedule by the boosted "
                                               "shard, `reference_rule_both_sides` keeps renderer.py:363's rule on both sides"},
            "gather_bytes_per_rank_at_8": int(-(-(-(-H * W // 128)) // 8) * 128 * 5 * 4),
            "note": "configs[3]-shaped (mip360/bonsai) 1080p inference frame, fixed eval model (seed 1234), device-resident loop, "
                    "through dist.render_frame_sharded (W = 1: no exchange); tests/test_gpu_frame1080.py checks the W = 2 / 8 "
                    "partitions against this frame"}


def frame_exchange(dev, H, W, reps=20):
    """what a rank of an 8-way sharded frame does besides rendering, timed on this one GPU (the reference's
    dormant `dist.all_gather(preds)` + index bookkeeping, nerf/utils.py:1555-1570): (a) gathering its own rays out of the frame
    (frame_plan `take`), (b) packing image / depth / weights into the [n/8, 5] block, (c) all_gather_into_tensor of that block on
    RCCL with a ONE-rank group (LAE_DIST_FORCE_COLLECTIVES: the launch and RCCL's own kernel, no wire), (d) the gather of the
    full-size [8 x n/8, 5] buffer into the caller's ray order (frame_plan `put`).  Median wall time of `reps` synchronised
    repetitions each."""
    import torch.distributed as dist
    from laenerf_amd import dist as D
    out = {}
    made_group = False
    try:
        if not dist.is_initialized():
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            import datetime
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                                    timeout=datetime.timedelta(seconds=60))
            made_group = True
        n = H * W
        plan = D.frame_plan(n, 0, 8, dev, (H, W))
        o = torch.rand(n, 3, device=dev); d = torch.rand(n, 3, device=dev)
        ns = plan["n_shard"]
        img, dep, ws = torch.rand(ns, 3, device=dev), torch.rand(ns, device=dev), torch.rand(ns, device=dev)
        big = torch.rand(8 * ns, 5, device=dev)

        def med(fn):
            ts = []
            for it in range(reps + 3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return sorted(ts[3:])[reps // 2] * 1e3
        out["take_rays_ms"] = round(med(lambda: (o[plan["take"]], d[plan["take"]])), 4)
        out["pack_block_ms"] = round(med(lambda: torch.cat([img, dep[:, None], ws[:, None]], dim=1)), 4)
        block = torch.cat([img, dep[:, None], ws[:, None]], dim=1)
        was = D.FORCE_COLLECTIVES
        D.FORCE_COLLECTIVES = True
        try:
            out["all_gather_w1_rccl_ms"] = round(med(lambda: D.exchange_blocks(block, 1)), 4)
        finally:
            D.FORCE_COLLECTIVES = was
        out["put_full_frame_ms"] = round(med(lambda: big[plan["put"]]), 4)
        # the sync + wall clock of one empty repetition is in every figure above: measure and subtract it once from the total
        empty = med(lambda: None)
        parts = [out["take_rays_ms"], out["pack_block_ms"], out["all_gather_w1_rccl_ms"], out["put_full_frame_ms"]]
        out["sync_overhead_ms"] = round(empty, 4)
        out["total_ms_per_frame"] = round(sum(max(p_ - empty, 0.0) for p_ in parts), 4)
        out["backend"] = dist.get_backend()
        out["note"] = "one-rank RCCL group on this GPU: launch + local copy of the collective, no xGMI transfer"
    except Exception as ex:                                      # a diagnostic must never take the line down
        out["error"] = repr(ex)
    finally:
        if made_group:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
    return out


def grid_update(dev):
    """occupancy-grid maintenance (update_extra_state, renderer.py:555-649; every 16 train steps, nerf/utils.py:1465) on a
    separate model of the same architecture: full sweeps (first 16 calls: 128^3 cells) and partial sweeps afterwards"""
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(3)
    net = NeRFNetwork(bound=1).to(dev)
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    r = NeRFRenderer(net, bound=1, density_thresh=10).to(dev)
    from laenerf_amd.optim import FusedAdam
    FusedAdam(net, param_groups=net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)   # as in training: the fp16 shadow table exists (no per-call cast)
    ts = []
    for it in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.float16):
            r.update_extra_state()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    full, part = sorted(ts[2:16])[7], sorted(ts[17:])[3]
    return {"full_sweep_ms": round(full, 3), "partial_sweep_ms": round(part, 3), "every_steps": 16,
            "amortised_us_per_step": round(part / 16 * 1e3, 1),
            "note": "not inside `value`: the reference runs it between train steps (nerf/utils.py:1465)"}


def style_step(dev, P=100000, steps=30, switches=None, G=8, pipeline=True):
    """configs[4]: one optimisation step of LAENeRF's palette network (train_LAENeRF_step, nerf/utils.py:980-1043, point-wise
    losses) on P region-masked points: hash-grid encode -> weight / offset MLPs -> palette recomposition -> MSE + weight +
    offset + palette losses -> backward -> Adam(lr 1e-3) under a GradScaler.  Synthetic x_term in a 0.3-radius ball
    (SURVEY.md 8d), 4 G point sets ("views": the reference's loader hands the step one view's points, editing/edit_dataset.py:
    236-260, known before the step starts).  `one_graph_per_step_ms`: one captured graph per step, replayed (rounds 2-4's figure);
    `ms_per_step` (round 5): the headline's grouped two-stream scheme -- the counting half of the hash-grid backward of the views of
    group k+2 (positions only, LAENeRF.plan_backward) replayed on a side stream beside the steps of group k."""
    from types import SimpleNamespace
    from laenerf_amd.editing import LAENeRF
    from laenerf_amd.optim import FusedAdam
    params = SimpleNamespace(bound=1, num_palette_bases=8, style_weight=0, weight_loss_uniform=1e-3, weight_loss_non_uniform=1e-3,
                             offset_loss=1e-2, palette_loss_valid=1.0, palette_loss_distinct=1e-2)
    torch.manual_seed(7)
    m = LAENeRF(params, dir_encoding="sphere_harmonics").to(dev)
    m.train()
    for k, val in (switches or {}).items():                    # A/B runs (tools/style_step_ab.py): fused_inputs / ffmlp_shadows off
        assert hasattr(m, k), k
        setattr(m, k, val)
    opt = FusedAdam(m, param_groups=m.get_params(1e-3), betas=(0.9, 0.999), eps=1e-8)
    views = []
    for _ in range(4 * G):
        v = torch.randn(P, 3, device=dev)
        views.append((v / v.norm(dim=-1, keepdim=True) * 0.3 * torch.rand(P, 1, device=dev) ** (1 / 3),
                      torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1), torch.rand(P, 3, device=dev)))

    def body(view, plan=None):
        x, d, target = view
        with torch.autocast("cuda", dtype=torch.float16):
            # recomposition + MSE + weight + offset + palette losses as one node (palette.hip)
            loss, pred, w, o = m.forward_train_loss(x, d, target, params, opt, with_palet_loss=True, plan=plan)
        opt.backward(loss)
        opt.step()
        return P
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            body(views[i], m.plan_backward(views[i][0]) if i else None)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(views[0])
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt_one = (time.perf_counter() - t0) / steps
    out = {"points": P, "one_graph_per_step_ms": round(dt_one * 1e3, 4)}
    # the same step on points as a view hands them over: termination points of neighbouring pixels lie next to each other on the
    # region's surface (EditDataset keeps a view's selected pixels in pixel order, editing/edit_dataset.py:147-160) -- here a
    # jittered row-major lattice on a spherical cap of the same 0.3 radius.  The hash-grid kernels run at the rate of their cache
    # hits, so the random ball above is the pessimistic end (SURVEY 8d's definition; it stays the quoted figure).
    side_n = int(P ** 0.5)
    Pc = side_n * side_n // 16 * 16
    u = torch.linspace(-0.9, 0.9, side_n, device=dev)
    uu, vv = torch.meshgrid(u, u, indexing="ij")
    cap = torch.stack([uu, vv, torch.sqrt((1.6 - uu * uu - vv * vv).clamp_min(0.0))], -1).reshape(-1, 3)[:Pc]
    xc = (torch.nn.functional.normalize(cap, dim=-1) * 0.3 + 1e-4 * torch.randn(Pc, 3, device=dev)).contiguous()
    coherent = (xc, torch.nn.functional.normalize(torch.randn(Pc, 3, device=dev), dim=-1), torch.rand(Pc, 3, device=dev))

    def body_c():
        with torch.autocast("cuda", dtype=torch.float16):
            loss, pred, w, o = m.forward_train_loss(*coherent, params, opt, with_palet_loss=True)
        opt.backward(loss)
        opt.step()
    with torch.cuda.stream(side):
        for i in range(3):
            body_c()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gc_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gc_):
        body_c()
    for _ in range(3):
        gc_.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        gc_.replay()
    torch.cuda.synchronize()
    out["surface_ordered_points"] = {"points": Pc, "one_graph_per_step_ms": round((time.perf_counter() - t0) / steps * 1e3, 4),
                                     "note": "pixel-ordered termination points on a spherical cap (what a view's batch looks like); not the quoted figure"}
    dt = dt_one
    if pipeline and m.ffmlp_shadows and m.plan_backward(views[0][0]) is not None:      # (A/B switches: tools/style_step_ab.py says why)
        step, _ = grouped_pipeline(None, opt, views, G, ahead_fn=lambda view: m.plan_backward(view[0]), step_fn=body)
        n = (steps + G - 1) // G * G
        for i in range(4 * G):
            step(i)
        wins = []
        step.host.update(s=0.0, groups=0)
        for w in range(3):                                     # three windows, the median counts: the first pipelined window of a process
            torch.cuda.synchronize(); t0 = time.perf_counter()   # that has replayed other graphs before can carry a one-off ~150 ms stall
            for i in range(n):
                step(4 * G + w * n + i)
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t0) / n)
        dt = sorted(wins)[1]
        out["steps_per_graph_replay"] = G
        out["windows_ms"] = [round(v * 1e3, 4) for v in wins]
        out["side_stream"] = SIDE_STREAM_PROBES[-1] if SIDE_STREAM_PROBES else None
        out["host_issue_ms_per_step"] = round(step.host["s"] / max(step.host["groups"], 1) / G * 1e3, 4)
    out.update({"ms_per_step": round(dt * 1e3, 4), "Mpoints_per_s": round(P / dt / 1e6, 2),
                "note": "LAENeRF palette network: encode + 2 MLPs + palette recomposition, fwd + bwd + Adam; HIP-graph replay, the counting "
                        "half of the grid backward of the views two groups ahead on a side stream (positions only), like the headline's march"})
    return out


def edit_extract(dev, n_views=8, batch_views=2):
    """configs[4] around its inner loop: the recolor flow is extraction -> 10 000 style steps -> distillation (scripts/run_mip360.sh:
    58-59); `style_step` times the middle, this times the extraction -- EditDataset.__init__'s per-view loop (editing/
    edit_dataset.py:74-234) as laenerf_amd.editing.extract_views over `n_views` bonsai-shaped 1080p poses at bound 2 with an edit
    grid AND a grow grid (--smooth_trans_weight: two device-resident distill renders per batch, the reference's selection rules,
    the nearest-grow-point distances, the crop terms).  `render_share`: the part of a view spent in get_rays + the two
    lae_render_frame calls; the rest is the selection / cdist / crop tail."""
    from laenerf_amd import raymarching, synthetic as S
    from laenerf_amd.editing import edit_dataset as ED
    net, r = eval_model(dev, bound=2, seed=1234)
    r.density_scale = 30.0                                     # a trained scene: opaque surfaces (weights saturate)
    H, W = 1080, 1920
    f = 1111.1 * H / 800
    intr = np.array([f, f, W / 2, H / 2], np.float32)
    poses = np.zeros((n_views, 4, 4), np.float32)
    for i in range(n_views):                                   # an orbit inside the bound-2 box, looking at the origin
        a = 2 * np.pi * i / n_views
        p = np.array([1.6 * np.cos(a), 1.6 * np.sin(a), 0.35 + 0.1 * np.sin(3 * a)])
        fwd = -p / np.linalg.norm(p)
        right = np.cross(np.array([0, 0, 1.0]), fwd); right /= np.linalg.norm(right)
        up = np.cross(fwd, right)
        poses[i, :3, 0], poses[i, :3, 1], poses[i, :3, 2], poses[i, :3, 3], poses[i, 3, 3] = right, up, fwd, p, 1
    poses = torch.from_numpy(poses).to(dev)
    dens = torch.from_numpy(S.flower_density_grid()).to(dev)                          # [2, 128^3], Morton order
    coords = raymarching.morton3D_invert(torch.arange(128 ** 3, dtype=torch.int32, device=dev))
    near_origin = ((coords.float() - 63.5).abs().amax(dim=1) < 20)                   # the edit region: a box around the centre ...
    ring = ((coords.float() - 63.5).abs().amax(dim=1) < 26) & ~near_origin           # ... the grow region: the shell around it
    edit = raymarching.packbits(torch.where(near_origin[None], dens, torch.zeros_like(dens)).contiguous(), 10.0)
    grow = raymarching.packbits(torch.where(ring[None], dens, torch.zeros_like(dens)).contiguous(), 10.0)
    images = torch.rand(n_views, H, W, 3, device=dev)

    def run_all():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        views, occluded = ED.extract_views(r, poses, intr, H, W, edit, images, batch_views=batch_views, grow_grid=grow)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, views, occluded

    def run_render_only():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i0 in range(0, n_views, batch_views):
            ED._render_views(r, poses[i0:i0 + batch_views], intr, H, W, edit, grow)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run_all()                                                  # warm-up (workspaces, caches)
    t_all, views, occluded = min((run_all() for _ in range(3)), key=lambda x: x[0])
    t_render = min(run_render_only() for _ in range(3))
    kept = [int(v["indices"].numel()) for v in views]
    interp = [int(v["indices_interp"].numel()) for v in views if "indices_interp" in v]
    return {"ms_per_view": round(t_all / n_views * 1e3, 2), "views": n_views, "batch_views": batch_views, "rays_per_view": H * W,
            "render_ms_per_view": round(t_render / n_views * 1e3, 2), "render_share": round(t_render / t_all, 3),
            "selection_cdist_crop_tail_ms_per_view": round((t_all - t_render) / n_views * 1e3, 2),
            "views_kept": len(views), "views_occluded": len(occluded), "rays_kept_per_view": int(np.mean(kept)) if kept else 0,
            "transition_pixels_per_view": int(np.mean(interp)) if interp else 0,
            "note": "laenerf_amd.editing.extract_views: get_rays + two device-resident distill renders (edit grid, grow grid) per batch of "
                    "views, then the reference's selection rules, nearest-grow-point distances and crop terms per view; fixed eval model "
                    "(seed 1234, density scale 30), bound 2, synthetic occupancy, results stay on the device"}


def grouped_pipeline(r, opt, batches, G, groups_ahead=2, ahead_fn=None, step_fn=None, side_priority=0):
    """the headline's execution scheme for any renderer / optimizer pair (fused criterion): per group of G resident ray batches
    two captured HIP graphs -- G x {ray/box, march, counting half of the grid backward} and G x {encoder, head, compositing +
    criterion, backward, Adam} -- the first replayed on a side stream `groups_ahead` groups before the second (it reads no
    weight).  Returns step(i) (i = consecutive step numbers; a replay is issued at every G-th) and the samples per batch.
    ahead_fn(batch) -> state / step_fn(batch, state) -> units: another pair of halves with the same property (the first reads no
    weight), e.g. the LAENeRF step's counting half of the grid backward and the step itself (style_step)."""
    n_batches = len(batches)
    assert n_batches % G == 0 and n_batches // G >= 2 * groups_ahead
    side_priority = int(os.environ.get("LAE_BENCH_SIDE_PRIO", side_priority))                # A/B switch
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream(priority=side_priority) if side_priority else concurrent_side_stream()[0]
    P = n_batches // G
    if ahead_fn is None:
        def ahead_fn(batch):
            return r.march_train(batch[0], batch[1], perturb=True, max_steps=1024, plan_backward=True)

        def step_fn(batch, state):
            with torch.autocast("cuda", dtype=torch.float16):
                res = r.shade_train(state, bg_color=1, gt=batch[2], scaler=opt)
            opt.backward(res["loss"])
            opt.step()
            return res["n_samples"]
    marched, n_samples, g_side, g_main = [], [], [], []
    side.wait_stream(main)
    for p_ in range(P):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=g_side[0].pool() if g_side else None):
            for b in range(p_ * G, (p_ + 1) * G):
                marched.append(ahead_fn(batches[b]))
        g_side.append(g)
    for p_ in range(P):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=g_main[0].pool() if g_main else None):
            for b in range(p_ * G, (p_ + 1) * G):
                n_samples.append(step_fn(batches[b], marched[b]))
        g_main.append(g)
    ev_side = [torch.cuda.Event() for _ in range(P)]
    ev_main = [torch.cuda.Event() for _ in range(P)]
    state = {"primed": -1}
    A = groups_ahead

    def launch_side(p_):
        with torch.cuda.stream(side):
            g_side[p_].replay()
            ev_side[p_].record(side)

    def step(i):
        b = i % n_batches
        if b % G:
            return n_samples[b]
        p_ = b // G
        t_host = time.perf_counter()
        if state["primed"] != p_:
            side.wait_stream(main)
            for a in range(A):
                launch_side((p_ + a) % P)
        if not ev_side[p_].query():
            main.wait_event(ev_side[p_])
        g_main[p_].replay()
        ev_main[p_].record(main)
        nxt = (p_ + A) % P
        side.wait_event(ev_main[nxt]) if i >= G else None
        launch_side(nxt)
        state["primed"] = (p_ + 1) % P
        host["s"] += time.perf_counter() - t_host            # what the host spends issuing one group (two graph launches + events):
        host["groups"] += 1                                  # above G x the step's GPU time the scheme is host-bound
        return n_samples[b]
    host = step.host = {"s": 0.0, "groups": 0}
    for p_ in range(P):
        ev_main[p_].record(main)
    return step, n_samples


def _pipelined_train_step(dev, bound, bitfield_np, make_rays, steps, n_rays, G, seed, with_operator_timing=True):
    """the headline's train step (march -> encode -> MLPs -> composite -> MSE -> backward -> Adam, grouped two-stream pipeline) on
    another scene: model of `bound`, occupancy `bitfield_np`, ray batches from make_rays(b).  Returns the timing dict pieces the
    `flower_step` / `sparse_step` objects are built from."""
    from laenerf_amd import backend
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.optim import FusedAdam
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(seed)
    net = NeRFNetwork(bound=bound).to(dev)
    r = NeRFRenderer(net, bound=bound, min_near=0.2).to(dev)
    r.density_bitfield = torch.from_numpy(bitfield_np).to(dev)
    opt = FusedAdam(net, param_groups=net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    batches = []
    for b in range(4 * G):
        o, d = make_rays(b)
        batches.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.rand(n_rays, 3, device=dev)))
    net.train()

    def body(b):
        o, d, gt = batches[b % len(batches)]
        with torch.autocast("cuda", dtype=torch.float16):
            res = r.render_train(o, d, bg_color=1, perturb=True, max_steps=1024, gt=gt, scaler=opt)
        opt.backward(res["loss"])
        opt.step()
        return res["n_samples"]
    for i in range(17):                                       # mean_count mode needs 16 sized steps (renderer.py:644-647)
        body(i)
    r.update_mean_count()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                             # warm-up on a capture-capable stream (allocations, workspaces)
        for i in range(3):
            body(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # one un-pipelined graph (the round-2 figure, for comparison)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt_one = (time.perf_counter() - t0) / steps
    step, n_samples = grouped_pipeline(r, opt, batches, G)
    steps = (steps + G - 1) // G * G
    for i in range(4 * G):
        step(i)
    wins = []
    for w in range(3):                                        # three windows, the median counts
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            step(4 * G + w * steps + i)
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t0) / steps)
    dt = sorted(wins)[1]
    tm = {}
    if with_operator_timing:
        with backend.kernel_timing(only=None) as kt:
            for i in range(10):
                body(i)
        tm = kt.result
    gf = tm.get("grid_encode_forward", {"ms": float("nan"), "units": 0, "calls": 0})
    per_launch = gf["units"] / max(gf["calls"], 1)
    us = gf["ms"] / max(gf["calls"], 1) * 1e3
    gbs = per_launch * GRID_FWD_BYTES_FP16 / (us * 1e-6) / 1e9 if gf["calls"] else float("nan")
    return {"ms_per_step": round(dt * 1e3, 4), "Mrays_per_s": round(n_rays / dt / 1e6, 3), "rays": n_rays,
            "samples_per_step": int(np.mean(n_samples)), "one_graph_per_step_ms": round(dt_one * 1e3, 4),
            "steps_per_graph_replay": G, "windows_ms": [round(v * 1e3, 4) for v in wins],
            "grid_forward": {"avg_launch_us": round(us, 2), "samples_per_launch": int(per_launch), "achieved_GBs": round(gbs, 1),
                             "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_sample": GRID_FWD_BYTES_FP16,
                             "table_entries": int(net.encoder.embeddings.shape[0]), "finest_resolution": 2048 * bound},
            "operator_ms_per_step": {k: round(v["ms"] / 10, 4) for k, v in sorted(tm.items())}}


def flower_step(dev, steps=40, n_rays=4096, G=8):
    """configs[2] (llff/flower: scripts/configs_llff/flower.sh -- bound 2 -> 2 cascades, offset (0, 0, 1.5): cameras inside
    the box, min_near 0.2, table of 6 328 848 entries with finest resolution 4096): the same train step as the headline
    (march -> encode -> MLPs -> composite -> MSE -> backward -> Adam) on forward-facing synthetic rays, through the headline's
    grouped two-stream pipeline (round 3; round 2 replayed one un-pipelined graph per step).  Extra field, not `value`; also
    reports the roofline figure of its hash-grid forward (HIP events around the call in 10 eager steps)."""
    from laenerf_amd import synthetic as S
    out = _pipelined_train_step(dev, 2, S.pack_bits_np(S.flower_density_grid(), 10.0), lambda b: S.flower_like_rays(n_rays, seed=5 + b),
                                steps, n_rays, G, seed=99)
    out["note"] = ("configs[2]-shaped (llff/flower): bound 2, 2 cascades, cameras inside the box; grouped two-stream pipeline like the "
                   "headline (march + counting pass of step k+2 beside shading / backward / Adam of step k)")
    return out


def sparse_step(dev, steps=40, n_rays=4096, G=8):
    """SURVEY 8d's second occupancy preset on the headline's pipeline: configs[1]-shaped model and cameras, "lego-like sparse"
    bitfield (laenerf_amd.synthetic.lego_sparse_density_grid: 3.05 % of the cells occupied, ~13 samples per ray against the
    headline's 13 % / ~63).  Extra field, not `value`: with a fifth of the samples the step is bound by its launch chain and the
    optimizer's pass over the table, not by the sample kernels."""
    from laenerf_amd import synthetic as S
    grid = S.lego_sparse_density_grid()
    out = _pipelined_train_step(dev, 1, S.pack_bits_np(grid, 10.0), lambda b: S.lego_like_rays(n_rays, seed=700 + b, n_views=1),
                                steps, n_rays, G, seed=77)
    out["occupied_fraction"] = round(float((grid > 10.0).mean()), 4)
    out["samples_per_ray"] = round(out["samples_per_step"] / n_rays, 1)
    out["note"] = ("configs[1] model and cameras on the sparse occupancy preset (3 % of the cells), same grouped two-stream pipeline, "
                   "optimizer inside; median of three windows")
    return out


class _SegmentProbe:
    """device time of an eager step that reads the device from the host in the middle: the stretch between two host reads is
    bracketed by a HIP-event pair behind a spin kernel long enough for the host to enqueue the whole stretch (so the events see
    kernels back to back, not the host's launch gaps); the step's device time is the sum of its stretches.
    tools/reference_chain.py calls cut() before each host read and begin() after it."""

    def __init__(self, sleep_cycles):
        self.sleep_cycles, self.pairs, self.open = int(sleep_cycles), [], None

    def begin(self):
        if self.open is None:
            torch.cuda._sleep(self.sleep_cycles)
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self.open = e0

    def cut(self):
        if self.open is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.pairs.append((self.open, e1))
            self.open = None

    def take_ms(self):
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        self.pairs = []
        return ms


def drop_in_step(dev, fused_ms, steps=40, n_rays=4096):
    """The train step of a LAENeRF checkout that has added INTEGRATION.md 1's three lines and changed nothing else: the reference's
    own operator sequence (nerf/renderer.py:259-333 -> nerf/network_ff.py:51-79 -> the wrappers' allocate-then-call rules ->
    torch MSE -> GradScaler -> torch.optim.Adam, nerf/utils.py:1472-1478), eager, through the four reference-named backend modules
    (tools/reference_chain.py).  Same rays, occupancy, table and MLP shapes as the headline.  `ms_per_step`: wall clock of
    K eager steps; `device_ms_per_step`: HIP events around the stretches between the step's host reads (two `torch.any` of
    network_ff.py:72, GradScaler's found_inf), each behind a spin kernel so that launch gaps do not count."""
    from laenerf_amd import synthetic as S
    from tools.reference_chain import ReferenceChain, drop_in_train_step
    out = {}
    for nan_check in (True, False):
        torch.manual_seed(1234)
        chain = ReferenceChain(bound=1, min_near=0.2, nan_check=nan_check).to(dev).train()
        chain.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
        opt = torch.optim.Adam(chain.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)       # main_nerf.py:223
        scaler = torch.amp.GradScaler("cuda")
        batches = []
        for b in range(16):
            o, d = S.lego_like_rays(n_rays, seed=b, n_views=1)
            batches.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.rand(n_rays, 3, device=dev)))
        for i in range(34):                                   # 16 host-sized steps, then mean_count mode (renderer.py:644-647)
            drop_in_train_step(chain, opt, scaler, batches[i % 16])
            if (i + 1) % 16 == 0:
                chain.update_mean_count()
        rows = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            _, res = drop_in_train_step(chain, opt, scaler, batches[i % 16])
            rows.append(res["n_rows"])
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        # device time: the same step with event pairs around its host-read-free stretches; unscale_ as its own call, so that the
        # stretch up to GradScaler's host read ends there and the optimizer's kernels start a new one (optimizer pre-step hook)
        probe = _SegmentProbe(sleep_cycles=8_000_000)         # ~4.5 ms of spin per stretch: > the host's enqueue time of any stretch
        hook = opt.register_step_pre_hook(lambda *a, **k: probe.begin())
        per_step = []
        import gc
        gc_was = gc.isenabled()
        gc.collect(); gc.disable()
        try:
            for i in range(12):
                probe.begin()
                drop_in_train_step(chain, opt, scaler, batches[i % 16], probe=probe, split_unscale=True)
                probe.cut()
                per_step.append(probe.take_ms())
        finally:
            hook.remove()
            if gc_was:
                gc.enable()
        per_step = per_step[2:]
        dev_ms = sorted(sum(p) for p in per_step)[len(per_step) // 2]
        key = "as_written" if nan_check else "without_nan_check"
        out[key] = {"ms_per_step": round(wall * 1e3, 4), "Mrays_per_s": round(n_rays / wall / 1e6, 3),
                    "device_ms_per_step": round(dev_ms, 4), "device_stretches_ms": [round(v, 4) for v in per_step[len(per_step) // 2]],
                    "bound_by": "host (launch issue + host reads)" if wall * 1e3 > 1.15 * dev_ms else "device",
                    "rows_per_step": int(np.mean(rows))}
    # INTEGRATION.md 3b's ladder of optional edits: FusedAdam in the place of torch.optim.Adam + GradScaler, nothing else; then also
    # `nerf_head` in the place of network_ff.py:57-79
    for key, fused_head in (("fused_adam_edit", False), ("fused_adam_and_head_edits", True)):
        try:
            from laenerf_amd.optim import FusedAdam
            from tools.reference_chain import one_edit_train_step
            torch.manual_seed(1234)
            chain = ReferenceChain(bound=1, min_near=0.2, nan_check=True, fused_head=fused_head).to(dev).train()
            chain.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
            fopt = FusedAdam(chain, param_groups=chain.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
            for i in range(34):
                one_edit_train_step(chain, fopt, batches[i % 16])
                if (i + 1) % 16 == 0:
                    chain.update_mean_count()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(steps):
                one_edit_train_step(chain, fopt, batches[i % 16])
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps
            probe = _SegmentProbe(sleep_cycles=8_000_000)
            per_step = []
            for i in range(12):
                probe.begin()
                one_edit_train_step(chain, fopt, batches[i % 16], probe=probe)
                probe.cut()
                per_step.append(probe.take_ms())
            dev_ms = sorted(sum(p) for p in per_step[2:])[5]
            out[key] = {"ms_per_step": round(wall * 1e3, 4), "Mrays_per_s": round(n_rays / wall / 1e6, 3), "device_ms_per_step": round(dev_ms, 4),
                        "edit": ("laenerf_amd.optim.FusedAdam instead of torch.optim.Adam + GradScaler" +
                                 (" and laenerf_amd.ffmlp.nerf_head instead of network_ff.py:57-79" if fused_head else "") +
                                 " (INTEGRATION.md 3b); the reference's renderer and operator wrappers unchanged")}
        except Exception as ex:                                 # a diagnostic must never take the line down
            out[key] = {"error": repr(ex)}
    aw = out["as_written"]
    res = {"ms_per_step": aw["ms_per_step"], "Mrays_per_s": aw["Mrays_per_s"], "device_ms_per_step": aw["device_ms_per_step"],
           "device_stretches_ms": aw["device_stretches_ms"], "bound_by": aw["bound_by"], "rows_per_step": aw["rows_per_step"],
           "without_network_nan_check": out["without_nan_check"], "with_one_edit": out["fused_adam_edit"], "with_two_edits": out["fused_adam_and_head_edits"],
           "fused_headline_ms_per_step": round(fused_ms, 4),
           "device_time_vs_fused_step": round(aw["device_ms_per_step"] / fused_ms, 2),
           "wall_time_vs_fused_step": round(aw["ms_per_step"] / fused_ms, 2), "rays": n_rays}
    lp = os.path.join(ROOT, "profiles", "drop_in_launches.json")        # kernel count of one step from the committed rocprofv3 trace
    if os.path.exists(lp):
        try:
            res["launches_per_step"] = json.load(open(lp))
        except Exception:
            pass
    res["note"] = ("zero-edit drop-in: the reference's operator sequence and wrapper rules (per-call table cast, [L,B,C] output + permute, "
                   "separate ffmlp / trunc_exp / SH / cat / composite launches, zero-filled buffers, torch MSE + GradScaler + torch.optim.Adam), "
                   "eager, through the modules install_as_reference_backends() registers; `value` is this repository's fused driver instead")
    return res


def cpu_baseline_cfg1(n_threads, budget_s=8.0):
    """the cfg1 train step -- `run()` path (renderer.py:128-256), 1024 rays x 512 uniform steps, L=4 hash grid,
    nn.Linear-shaped nets (network.py:95-124) -- on the host cores: the oracle's (sequential C) operators for the hash grid and the
    SH encoder dealt over n_threads Python threads in chunks of points (ctypes releases the GIL), torch CPU tensors with
    torch.set_num_threads(n_threads) for everything else (the GEMMs, exp / sigmoid, cumprod, the reverse cumulative sums of the
    backward) -- round 3 ran those in numpy, which torch.set_num_threads does not reach.  Forward + backward,
    whole steps repeated until the budget is spent."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from laenerf_amd import synthetic as S
    offsets, pls = O.grid_offsets(num_levels=4, desired_resolution=2048)
    rng = np.random.default_rng(0)
    table = rng.uniform(-1e-4, 1e-4, (int(offsets[-1]), 2)).astype(np.float32)
    Wn = [rng.uniform(-0.3, 0.3, sh).astype(np.float32) for sh in ((64, 8), (16, 64), (64, 31), (64, 64), (3, 64))]
    o, d = S.lego_like_rays(1024, H=64, W=64, focal=1111.1 * 64 / 800, seed=3)
    N, T = 1024, 512
    torch.set_num_threads(n_threads)
    pool = ThreadPoolExecutor(n_threads)

    def chunks(*arrays):
        idx = np.array_split(np.arange(arrays[0].shape[0]), n_threads)
        return [tuple(a[i] for a in arrays) for i in idx if i.size]
    W = [torch.from_numpy(w) for w in Wn]
    ot, dt_ = torch.from_numpy(o), torch.from_numpy(d)
    lin = torch.linspace(0, 1, T)

    def step():
        nears, fars = O.near_far_from_aabb(o, d, [-1, -1, -1, 1, 1, 1], 0.2)
        nears, fars = torch.from_numpy(nears), torch.from_numpy(fars)
        z = nears[:, None] + (fars - nears)[:, None] * lin[None]
        xyz = torch.clamp(ot[:, None] + dt_[:, None] * z[..., None], -1, 1).reshape(-1, 3)
        x01 = ((xyz + 1) / 2).contiguous().numpy()
        enc = np.concatenate(list(pool.map(lambda a: O.grid_encode_forward(a[0], table, offsets, pls, 16, out_blc=True)[0], chunks(x01))))
        enc = torch.from_numpy(enc)
        h1 = torch.relu(enc @ W[0].T); h = h1 @ W[1].T
        sigma = torch.exp(h[:, 0]).reshape(N, T)
        deltas = torch.cat([z[:, 1:] - z[:, :-1], ((fars - nears) / T)[:, None]], 1)
        alphas = 1 - torch.exp(-deltas * sigma)
        trans = torch.cumprod(torch.cat([torch.ones(N, 1), 1 - alphas + 1e-15], 1), 1)[:, :-1]
        w = alphas * trans
        sh = np.concatenate(list(pool.map(lambda a: O.sh_encode_forward(a[0], 4)[0], chunks(np.repeat(d, T, axis=0)))))
        cin = torch.cat([torch.from_numpy(sh), h[:, 1:]], 1)
        c1 = torch.relu(cin @ W[2].T); c2 = torch.relu(c1 @ W[3].T); c3 = c2 @ W[4].T
        rgb = torch.sigmoid(c3).reshape(N, T, 3)
        image = (w[..., None] * rgb).sum(1) + (1 - w.sum(1))[:, None]
        # backward (MSE against 0.5): image -> rgb, w -> sigma (reverse cumulative sums) -> nets -> table
        gimg = 2 * (image - 0.5) / image.numel()
        grgb = (w[..., None] * gimg[:, None]).reshape(-1, 3)
        gw = ((rgb - 1.0) * gimg[:, None]).sum(-1)
        gww = gw * w
        sfx = torch.flip(torch.cumsum(torch.flip(gww, [1]), 1), [1]) - gww
        gsigma = (deltas * (gw * trans * (1 - alphas) - sfx)).reshape(-1)
        r2 = rgb.reshape(-1, 3)
        gc3 = grgb * (r2 * (1 - r2))
        gc2 = (gc3 @ W[4]) * (c2 > 0); gc1 = (gc2 @ W[3]) * (c1 > 0); gcin = gc1 @ W[2]
        gh = torch.cat([(gsigma * sigma.reshape(-1))[:, None], gcin[:, 16:]], 1)
        genc = ((gh @ W[1]) * (h1 > 0)) @ W[0]
        gW = [((gh @ W[1]) * (h1 > 0)).T @ enc, h1.T @ gh, gc1.T @ cin, gc2.T @ c1, gc3.T @ c2]     # weight gradients of the five GEMMs
        assert len(gW) == 5
        list(pool.map(lambda a: O.grid_encode_backward(a[0], a[1], table.shape, offsets, pls, 16, grad_blc=True), chunks(genc.contiguous().numpy(), x01)))
        return N
    step()                                                    # warm-up (page-in, thread pools)
    t0 = time.perf_counter(); n = 0; k = 0
    while time.perf_counter() - t0 < budget_s or k < 2:
        n += step(); k += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt / 1e6, 6), "unit": "Mrays/s", "ms_per_step": round(dt / k * 1e3, 1), "steps": k, "cores": n_threads,
            "sample": f"{k} cfg1 train steps (run() path: 1024 rays x 512 steps = 524288 points, L=4 grid, nn.Linear-shaped nets), forward + "
                      f"backward incl. weight gradients, oracle operators for the hash grid / SH in {n_threads} chunks + torch CPU ops on {n_threads} threads, {dt:.1f} s"}


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside any rendezvous environment: start the N ranks ourselves, as a FRESH child --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>`
    -- before this process has made any GPU call (subprocess, never exec), relay the child's stdout (rank 0's one JSON line) and
    return its exit code; non-zero too when the child printed no line carrying `n_gpus: N`.  The driver's own torchrun form sets
    WORLD_SIZE and never comes here."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()                       # counts devices without initialising the GPU on this image
    if n_dev < args.gpus and os.environ.get("LAE_BENCH_SINGLE_DEVICE") != "1":
        print(f"bench.py: --gpus {args.gpus} asks for {args.gpus} ranks on {args.gpus} GPUs of this node, which shows {n_dev}", file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    sys.stdout.write(child.stdout)
    sys.stdout.flush()
    if child.returncode != 0:
        return child.returncode
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    try:
        ok = len(lines) == 1 and json.loads(lines[0]).get("n_gpus") == args.gpus
    except ValueError:
        ok = False
    if not ok:
        print(f"bench.py: the {args.gpus}-rank child did not print one line with n_gpus == {args.gpus}", file=sys.stderr)
        return 3
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                                   # no GPU call has happened in this process yet
            raise SystemExit(launch_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        # a line that says n_gpus = WORLD_SIZE under a command that asked for --gpus N would be a silent non-measurement
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}: pass the same number to both",
              file=sys.stderr)
        raise SystemExit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (one-GPU boxes): LAE_BENCH_DIST_BACKEND=gloo LAE_BENCH_SINGLE_DEVICE=1 runs every rank on cuda:0 over gloo,
    # which exercises the multi-rank control flow (build barrier, timing barriers, max-over-ranks) without RCCL
    backend_name = os.environ.get("LAE_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("LAE_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend_name)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from laenerf_amd import backend, build, synthetic as S
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()

    if args.workload == "flower":
        print(json.dumps({"flower_step": flower_step(dev)}), flush=True)
        return
    if args.workload == "frame1080":
        if args.steps == 200 and args.warmup == 40:            # the train defaults are too long for whole frames
            args.steps, args.warmup = 20, 3
        return frame_workload(args, world, rank, dev, backend_name)

    torch.manual_seed(1234 + rank)
    net = NeRFNetwork(bound=1).to(dev)                        # L=16, T=2^19, F=2; FFMLP 2x64 / 3x64
    r = NeRFRenderer(net, bound=1, min_near=0.2).to(dev)
    r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
    # main_nerf.py:223 + nerf/utils.py:1474-1482: Adam(betas=(0.9, 0.99), eps=1e-15) under a GradScaler.  Default: the
    # same arithmetic as three HIP kernels that also keep the encoder's fp16 table and gradient buffer (optim.py).
    if args.torch_optimizer:
        opt = torch.optim.Adam(net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=not args.no_graph)
        scaler = torch.amp.GradScaler("cuda")
    else:
        from laenerf_amd.optim import FusedAdam
        opt = scaler = FusedAdam(net, param_groups=net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    fused_loss = not args.torch_loss and not args.torch_optimizer
    spg = int(args.steps_per_graph)
    if spg <= 0:                                                   # largest divisor of --steps that is <= 16
        spg = max(dv for dv in range(1, 17) if args.steps % dv == 0)
    spg = max(1, min(spg, 16))
    while spg > 1 and args.steps % spg:
        spg -= 1
    n_batches = spg * max(4, -(-16 // spg))                        # resident ray batches: >= 4 groups of spg steps, >= 16 batches
    batches = []
    for b in range(n_batches):
        # one training view per step, random pixels of it: what the reference's loader does (DataLoader batch_size = 1,
        # nerf/provider.py:349; get_rays draws randint(0, H*W) pixel indices of that pose, nerf/utils.py:108)
        o, d = S.lego_like_rays(args.rays, seed=1000 * rank + b, n_views=1)
        batches.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev),
                        torch.rand(args.rays, 3, device=dev)))
    net.train()

    def step_body(o, d, gt):
        with torch.autocast("cuda", dtype=torch.float16):
            if fused_loss:                                  # criterion + GradScaler.scale inside the compositing op
                res = r.render_train(o, d, bg_color=1, perturb=True, max_steps=1024, gt=gt, scaler=scaler)
                loss = res["loss"]
            else:
                res = r.render_train(o, d, bg_color=1, perturb=True, max_steps=1024)
                loss = scaler.scale(torch.nn.functional.mse_loss(res["image"], gt))
        scaler.backward(loss) if fused_loss else loss.backward()
        if not args.no_optimizer and not args.dp:
            if args.torch_optimizer:
                scaler.step(opt)
                scaler.update()
            else:
                opt.step()
        return res["n_samples"]

    def dp_tail():                                          # --dp: outside the captured graph (a collective + 3 launches)
        if world > 1:
            from laenerf_amd.dist import allreduce_gradients
            allreduce_gradients(opt, world)
        opt.step()

    def zero_grad():
        if args.torch_optimizer:
            opt.zero_grad(set_to_none=True)
        elif args.no_optimizer:
            opt.zero_grad()                                # FusedAdam.step() zeroes what it consumes

    def step(i):
        o, d, gt = batches[i % n_batches]
        zero_grad()
        n = step_body(o, d, gt)
        if args.dp:
            dp_tail()
        return n

    # warm-up: the first 16 steps run in the reference's "mean_count <= 0" mode (sized by a D2H read), then the
    # running mean is refreshed every 16 steps exactly like update_extra_state does (renderer.py:644-647)
    n_warm = max(args.warmup, 17)
    for i in range(n_warm):
        step(i)
        if (i + 1) % 16 == 0:
            r.update_mean_count()
    r.update_mean_count() if r.local_step > 0 else None
    samples = []

    # Steady state has static shapes (sample buffers are sized by mean_count, renderer.py:644-646), no host sync,
    # a fused capturable Adam and device-side GradScaler, so the ~100 launches of one step are captured once into a
    # HIP graph and replayed: the eager step is bound by host launch overhead (~15 us per launch), not by the GPU.
    # One graph per resident ray batch (the batches already live in HBM, so a replay reads them in place); the graphs
    # share one memory pool because they never run concurrently.
    graph = None
    if args.dp and (args.torch_optimizer or args.no_optimizer):
        raise SystemExit("--dp needs the FusedAdam path")
    pipelined = not args.no_graph and not args.no_pipeline and not args.torch_optimizer and not args.no_optimizer and not args.dp
    if not args.no_graph:
        side, side_probe = concurrent_side_stream() if pipelined else (torch.cuda.Stream(), None)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on the capture stream (allocations, workspaces)
            for _ in range(3):
                zero_grad()
                step_body(*batches[0])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    if not args.no_graph and not pipelined:
        graphs, n_graph_samples = [], []
        for b in range(n_batches):
            zero_grad()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None):
                n_graph_samples.append(step_body(*batches[b]))
            graphs.append(g)
        graph = graphs[0]

        def step(i):                                        # noqa: F811  (replaces the eager step)
            graphs[i % n_batches].replay()
            if args.dp:
                dp_tail()
            return n_graph_samples[i % n_batches]
        for i in range(5):
            step(i)
    if pipelined:
        # The march reads rays and the occupancy bitfield only (no network weight), so the march of step k+1 is captured
        # as its own graph and replayed on a side stream while the main stream runs shading, backward and the optimizer
        # of step k.  The counting half of the hash-grid backward (positions only) rides along in that graph.  Same kernels and the same work per step; the two graph families use separate memory pools because
        # they run concurrently, and every march graph keeps its own output buffers (read by its shading graph).
        main = torch.cuda.current_stream()
        marched, n_graph_samples = [], []
        G = spg                                             # divides --steps and n_batches by construction
        if G > 1:
            # G consecutive steps per replay: {march(b) ... march(b+G-1)} on the side stream, one group ahead of
            # {shade, backward, Adam of b; ... of b+G-1} on the main stream.  Every step does exactly the kernels of the
            # one-step graphs; only the graph boundaries (15-18 us each, profiles/r1m) are shared by G steps.
            P = n_batches // G
            g_side = []
            for p_ in range(P):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=g_side[0].pool() if g_side else None):
                    for b in range(p_ * G, (p_ + 1) * G):
                        marched.append(r.march_train(batches[b][0], batches[b][1], perturb=True, max_steps=1024, plan_backward=True))
                g_side.append(g)
            g_main = []
            for p_ in range(P):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=g_main[0].pool() if g_main else None):
                    for b in range(p_ * G, (p_ + 1) * G):
                        with torch.autocast("cuda", dtype=torch.float16):
                            if fused_loss:
                                res = r.shade_train(marched[b], bg_color=1, gt=batches[b][2], scaler=scaler)
                                loss = res["loss"]
                            else:
                                res = r.shade_train(marched[b], bg_color=1)
                                loss = scaler.scale(torch.nn.functional.mse_loss(res["image"], batches[b][2]))
                        scaler.backward(loss) if fused_loss else loss.backward()
                        opt.step()
                        n_graph_samples.append(res["n_samples"])
                g_main.append(g)
            graph = g_main[0]
            ev_side = [torch.cuda.Event() for _ in range(P)]
            ev_main = [torch.cuda.Event() for _ in range(P)]
            state = {"primed": -1}

            def launch_side(p_):
                with torch.cuda.stream(side):
                    g_side[p_].replay()
                    ev_side[p_].record(side)

            A = 2 if P >= 4 else 1                            # side groups kept this far ahead of the main stream

            def step(i):                                    # noqa: F811
                b = i % n_batches
                if b % G:                                   # this step rides in the replay issued at the group's first step
                    return n_graph_samples[b]
                p_ = b // G
                if state["primed"] != p_:                   # first group after a break in the sequence: marches in line
                    side.wait_stream(main)
                    for a in range(A):
                        launch_side((p_ + a) % P)
                # With the side stream two groups ahead, the marches of this group finished while the host was still
                # enqueuing the previous group, so the main stream usually needs NO cross-stream wait: its graphs queue up
                # back to back.  (A wait on a not yet signalled event of another stream costs ~50 us at every boundary.)
                if not ev_side[p_].query():
                    main.wait_event(ev_side[p_])
                g_main[p_].replay()
                ev_main[p_].record(main)
                nxt = (p_ + A) % P
                # buffers of group `nxt` were last read by the main graph of that group one cycle ago (its event still holds
                # that record: this cycle's replay of it is not enqueued yet)
                side.wait_event(ev_main[nxt]) if i >= G else None
                launch_side(nxt)
                state["primed"] = (p_ + 1) % P
                return n_graph_samples[b]
            for p_ in range(P):
                ev_main[p_].record(main)
            # whole groups, ending right before the first timed step.  Six groups of replays (48 steps, ~20 ms): on a fresh
            # box the first ~10 ms of back-to-back replays run slower than the rest (first window 0.420 ms against 0.408 for
            # the next four with two groups), and K = 20 timed steps are only 8 ms
            WARM_GROUPS = 6
            first = ((n_warm + G - 1) // G) * G - 2 * G
            n_warm = first + WARM_GROUPS * G
            for i in range(first, n_warm):
                step(i)
            torch.cuda.synchronize()
    if pipelined and G == 1:
        g_march = []
        for b in range(n_batches):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=g_march[0].pool() if g_march else None):
                marched.append(r.march_train(batches[b][0], batches[b][1], perturb=True, max_steps=1024, plan_backward=True))
            g_march.append(g)
        # The rest of the step is two graphs, {encoder, head, compositing + criterion} and {backward, Adam}: the march of the
        # next step starts beside the SECOND one (LDS- / HBM-bound kernels with idle VALU slots) -- beside the first it
        # competed with the L2-request-bound encoder for the same wave slots (--march-beside forward: 0.503 ms/step).
        g_fwd, g_bwd, losses = [], [], []
        late = args.march_beside == "backward"
        for b in range(n_batches):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=g_fwd[0].pool() if g_fwd else None):
                with torch.autocast("cuda", dtype=torch.float16):
                    if fused_loss:
                        res = r.shade_train(marched[b], bg_color=1, gt=batches[b][2], scaler=scaler)
                        loss = res["loss"]
                    else:
                        res = r.shade_train(marched[b], bg_color=1)
                        loss = scaler.scale(torch.nn.functional.mse_loss(res["image"], batches[b][2]))
                n_graph_samples.append(res["n_samples"])
            g_fwd.append(g); losses.append(loss)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=g_fwd[0].pool()):
                scaler.backward(losses[b]) if fused_loss else losses[b].backward()
                opt.step()
            g_bwd.append(g)
        graph = g_fwd[0]
        ev_march = [torch.cuda.Event() for _ in range(n_batches)]
        ev_mid = [torch.cuda.Event() for _ in range(n_batches)]
        ev_rest = [torch.cuda.Event() for _ in range(n_batches)]
        state = {"primed": -1}

        def launch_march(b):
            with torch.cuda.stream(side):
                g_march[b].replay()
                ev_march[b].record(side)

        def step(i):                                        # noqa: F811
            b, nxt = i % n_batches, (i + 1) % n_batches
            if state["primed"] != b:                        # first step after a break in the sequence: march in line
                side.wait_stream(main)
                launch_march(b)
            main.wait_event(ev_march[b])
            g_fwd[b].replay()                               # main first, see the grouped path
            ev_mid[b].record(main)
            g_bwd[b].replay()
            ev_rest[b].record(main)
            side.wait_event(ev_mid[b] if late else ev_rest[(b - 1) % n_batches]) if (late or i > 0) else None
            launch_march(nxt)
            state["primed"] = nxt
            return n_graph_samples[b]
        for b in range(n_batches):
            ev_rest[b].record(main)
        for i in range(n_warm - 5, n_warm):                 # ends with the march of the first timed step in flight
            step(i)
        torch.cuda.synchronize()
    backend.enable_kernel_timing(not graph)                 # (--no-graph only: events inside the timed region; switched off right after it)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        samples.append(step(n_warm + i))
    sync_all()
    dt = time.perf_counter() - t0
    # four more windows of the same K steps right after the first (a 7 ms window drifts by a few % on a box); every window is
    # exactly K steps between barriers, `value` is their median
    windows = [dt]
    for wdw in range(1, 5):
        sync_all()
        tw = time.perf_counter()
        for i in range(args.steps):
            step(n_warm + wdw * args.steps + i)
        sync_all()
        windows.append(time.perf_counter() - tw)
    timing_grid = backend.collect_kernel_timing() if not graph else {}
    backend.enable_kernel_timing(False)
    # diagnostic (outside the timed region): per-operator device time of 20 eager steps (HIP events around each
    # operator on the launch stream); with graph replay this is also where the roofline kernel is timed, because
    # events cannot be read back from inside a replayed graph
    if graph:
        for i in range(3):                                  # untimed: the first eager steps after the replays (allocations, caches)
            o, d, gt = batches[i % n_batches]
            zero_grad()
            step_body(o, d, gt)
        torch.cuda.synchronize()
    # (the collector is off while the timing is on -- backend.enable_kernel_timing says why; `launch_us_min_median_max` shows the spread)
    n_diag = 20
    with backend.kernel_timing(only=None) as kt:
        for i in range(n_diag):
            o, d, gt = batches[i % n_batches]
            zero_grad()
            step_body(o, d, gt)
    timing_all = kt.result
    # the same 20 steps with the hash-grid backward in its two halves (counting pass + scans = what the side stream runs ahead;
    # fill + accumulate = what stays on the main stream), HIP events around each: the `roofline_more` objects below
    if fused_loss and not args.no_optimizer and not args.dp:
        with backend.kernel_timing(only=None) as kt:
            for i in range(n_diag):
                o, d, gt = batches[i % n_batches]
                zero_grad()
                with torch.autocast("cuda", dtype=torch.float16):
                    marched = r.march_train(o, d, perturb=True, max_steps=1024, plan_backward=True)
                    res = r.shade_train(marched, bg_color=1, gt=gt, scaler=scaler)
                scaler.backward(res["loss"])
                opt.step()
        timing_split = kt.result
    else:
        timing_split = {}
    if graph:
        timing_grid = {"grid_encode_forward": timing_all.get("grid_encode_forward", {"ms": float("nan"), "units": 0, "calls": 0})}
    # N > 1: the north star's split, beside the replica `value` -- the configs[3] frame ray-sharded over the ranks with ONE
    # all-gather per frame on the job's backend (RCCL under the driver), timed like `--workload frame1080`, and the same frame on
    # rank 0 alone for the speed-up.  Collective: every rank takes part, rank 0 reports.
    sharded = None
    if world > 1 and not args.no_frame:
        sharded = sharded_frame1080(world, rank, dev, backend_name, steps=10, warmup=2, with_n1=True)

    if world > 1:                                           # every window: max over ranks (all ranks call the collective)
        wmax = torch.tensor(windows, device=dev if backend_name == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
        windows = [float(x) for x in wmax.tolist()]
    first_window = windows[0]
    dt = sorted(windows)[len(windows) // 2]                 # `value` = the MEDIAN of the five K-step windows
    if rank == 0:
        ms = dt / args.steps * 1e3
        gf = timing_grid.get("grid_encode_forward", {"ms": float("nan"), "units": 0, "calls": 0})
        per_launch_bytes = gf["units"] / max(gf["calls"], 1) * GRID_FWD_BYTES_FP16
        achieved = per_launch_bytes / (gf["ms"] / max(gf["calls"], 1) * 1e-3) / 1e9 if gf["calls"] else float("nan")
        traffic, traffic_source = None, None
        tp = os.path.join(ROOT, "profiles", "grid_fwd_traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                traffic = tj.get("hbm_bytes_per_launch")
                # NOT a counter of this run: PMC passes need rocprofv3 around the process; the committed result of the most recent
                # counter run of this same command is read back (its file says which run)
                traffic_source = "read back from profiles/grid_fwd_traffic.json (%s), not counted in this run" % tj.get("source", tj.get("note", "committed PMC run"))
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s, lego 800x800 train-step (4096 rays/step)",
            "value": round(world * args.rays * args.steps / dt / 1e6, 4),
            "unit": "Mrays/s",
            "n_gpus": world, "world_size": (dist.get_world_size() if world > 1 else 1),
            "backend": (dist.get_backend() if world > 1 else None), "steps": args.steps, "warmup": n_warm,
            "warmup_requested": args.warmup,               # `warmup` = the untimed steps actually run (>= 17 for mean_count mode + whole replay groups)
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 (table, MLP) / f32 (march, composite)", "data": "synthetic",
            "config": {"workload": "configs[1]: lego-like 800x800 pinhole rays, 4096 random pixels of one view per step, L=16 T=2^19 F=2 hash grid "
                                   "+ 2x64 / 3x64 ffmlp, cascade 1, analytic occupancy (13% occupied), "
                                   "steady-state train step incl. backward + Adam",
                       "rays_per_step": args.rays, "samples_per_step": int(np.mean(samples)),
                       "optimizer_in_timed_region": not args.no_optimizer,
                       "occupancy_grid_maintenance_in_timed_region": False,
                       "hip_graph_replay": bool(graph),
                       "march_pipelined_on_side_stream": bool(pipelined), "steps_per_graph_replay": (G if pipelined else 1),
                       "parallelism": (f"{world} data-parallel ranks, one flat gradient all-reduce per dtype per step" if args.dp else
                                       f"{world} independent ray-batch replicas (no data-path collective)")},
            "roofline": {"kernel": "k_grid_fwd_lean (hash-grid encode forward, fp16 table)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_sample": GRID_FWD_BYTES_FP16,
                         "samples_per_launch": int(gf["units"] / max(gf["calls"], 1)),
                         "avg_launch_us": round(gf["ms"] / max(gf["calls"], 1) * 1e3, 2),
                         "launch_us_min_median_max": [round(gf.get(k, float("nan")) * 1e3, 2) for k in ("min_ms", "median_ms", "max_ms")],
                         "algorithmic_bytes_per_launch": int(per_launch_bytes),
                         "frac_of_peak_on_measured_bytes": (round(traffic / (gf["ms"] / max(gf["calls"], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                            if traffic and gf["calls"] else None),
                         "limiter": "on-chip: the 2 MiB level tables stay in the XCDs' L2s (measured traffic < algorithmic bytes); fine "
                                    "levels run at the L2 request rate, coarse levels at vector-memory issue (DESIGN.md 4)",
                         "timed": "HIP events on the launch stream around the call in 20 eager steps after the timed region"},
            "roofline_more": roofline_more(timing_split, dev, with_frame=not args.no_frame) if timing_split else None,
            "windows": {"ms_per_step": [round(wd / args.steps * 1e3, 4) for wd in windows],
                        "min": round(min(windows) / args.steps * 1e3, 4), "median": round(sorted(windows)[len(windows) // 2] / args.steps * 1e3, 4),
                        "max": round(max(windows) / args.steps * 1e3, 4), "first": round(first_window / args.steps * 1e3, 4),
                        "note": "5 consecutive windows of exactly K steps, each between barriers + synchronize, max over ranks; "
                                "`value` / `ms_per_step` are the MEDIAN window"},
            "side_stream": side_probe if not args.no_graph else None,        # concurrent_side_stream(): does the side stream run beside the main one?
            "operator_ms_per_step": {k: round(v["ms"] / n_diag, 4) for k, v in sorted(timing_all.items())},
        }
        if sharded is not None:
            sharded["note"] = ("configs[3]-shaped (mip360/bonsai) 1080p inference frame, rays sharded over the ranks of THIS job "
                               "(laenerf_amd/dist.py: 8x8-pixel-tile order, 128-ray units round-robin, one all_gather_into_tensor per frame), 10 frames between "
                               "barriers, max over ranks; n1_ms_per_frame: the whole frame on rank 0 alone on the reference schedule, "
                               "best_n1_ms_per_frame: on the fastest of {reference rule, 2N, 4N, 8N rows per iteration}; speedup_vs_n1 divides THAT by ms_per_frame")
            out["frame1080"] = sharded                         # the north star's 8-GPU split (not `value`)
        def extra(fn, *a, **k):                              # the objects beside `value`: a failure inside one of them must not take the line down (it shows as {"error": ...})
            try:
                return fn(*a, **k)
            except Exception as ex:
                return {"error": repr(ex)}
        if world == 1 and not args.no_dropin:
            out["drop_in_step"] = extra(drop_in_step, dev, ms)   # the reference's own operator sequence on the installed backends (not `value`)
        if world == 1 and not args.no_frame:
            out["eval_frame"] = extra(eval_frame, dev)              # the "ms/frame" half of BASELINE.json's metric (not `value`)
            out["eval_frame_surface"] = extra(eval_frame, dev, density_scale=30.0)   # the same frame on a trained-shaped density
            out["frame1080"] = extra(frame1080, dev)               # configs[3] on one GPU (not `value`)
        if world == 1 and not args.no_style:
            out["style_step"] = extra(style_step, dev)              # configs[4] inner loop (not `value`)
            out["edit_extract"] = extra(edit_extract, dev)          # configs[4] extraction around it (not `value`)
            out["grid_update"] = extra(grid_update, dev)
            out["flower_step"] = extra(flower_step, dev)            # configs[2]-shaped train step (not `value`)
            out["sparse_step"] = extra(sparse_step, dev)       # the sparse occupancy preset of SURVEY 8d (not `value`)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_rays, host_threads())
            out["cpu_baseline"]["cfg1_run_path"] = cpu_baseline_cfg1(host_threads())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
