"""-m gpu: BASELINE configs[3] -- mip360/bonsai-shaped 1920x1080 frame (scripts/configs_mip360/bonsai.sh: bound 2 -> 2 cascades,
512 KiB bitfield, 6 328 848-entry table, camera inside the box), rays sharded in 128-ray tiles (laenerf_amd/dist.py; the
reference's dormant all_gather(preds), nerf/utils.py:1555-1570) around the inference loop of nerf/renderer.py:335-387.

What can be bit-identical and what cannot: the reference's loop takes `n_step = max(min(N // n_alive, 8), 1)` samples per
ray and iteration, N = the rays of THE CALL, and `rays_t` is re-rounded at every iteration boundary (composite_rays adds the
deltas up again, raymarching.cu:985-1034).  A shard is a call with N / W rays, so its boundaries fall elsewhere and a ray's
result may move by rounding -- under the reference's own loop as much as here.  With one sample per iteration (max_n_step=1)
a ray's arithmetic does not depend on its neighbours: there the W-way partitions must reproduce the whole frame BIT FOR BIT,
which pins the sharding machinery (tile deal, padding rows, block layout, de-interleave) exactly; on the reference schedule
the partitions are asserted within 1e-5 (north_star: 1e-4 RGB)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from gpu_util import DEV, N, T

pytestmark = pytest.mark.gpu

# HIP frame loop against the ORACLE's loop (fp16 MLPs on both sides; what differs is the fp32 accumulate of the MFMA against the
# oracle FFMLP's and __expf against expf): bounds = ~2x the deviation observed on the MI355X box (printed by the tests), all well
# inside north_star's 1e-4 RGB -- rounds 1-3 asserted 3e-3 here
ORACLE_LOOP_TOL = {"weights_sum": 2e-6, "image": 2.5e-5, "depth": 1.5e-6}      # observed 9.5e-7 / 1.2e-5 / 5.7e-7

H, W = 1080, 1920


@pytest.fixture(scope="module")
def bonsai():
    import bench
    from laenerf_amd import synthetic as S
    net, r = bench.eval_model(torch.device(DEV), bound=2, seed=1234)
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)            # densities over orders of magnitude: rays end at different depths
    assert r.cascade == 2 and r.density_bitfield.numel() == 2 * 128 ** 3 // 8 and net.encoder.embeddings.shape[0] == 6328848
    o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
    return net, r, T(o), T(d)


def _bits(t):
    return t.contiguous().view(torch.int32)


def _render(r, **kw):
    def fn(ro, rd):
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            return r.render_eval(ro, rd, bg_color=1, max_steps=1024, **kw)
    return fn


def test_frame1080_partitions_reproduce_the_whole_frame_bit_for_bit(bonsai):
    from laenerf_amd.dist import assemble_frame, render_frame_sharded, render_shard
    net, r, o, d = bonsai
    fn = _render(r, max_n_step=1)
    whole = fn(o, d)
    assert float((whole["weights_sum"] > 0).float().mean()) > 0.2
    one = render_frame_sharded(fn, o, d, 0, 1)                 # W = 1 through the public entry point
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(_bits(one[k]), _bits(whole[k])), k
    tiled = render_frame_sharded(fn, o, d, 0, 1, image_hw=(H, W))   # rays dealt and rendered in 8 x 8 pixel tiles, results in the caller's order
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(_bits(tiled[k]), _bits(whole[k])), k
    from laenerf_amd.dist import pixel_tile_order
    order = pixel_tile_order((H, W), o.device)
    blocks = [render_shard(fn, o[order[0]], d[order[0]], rank, 8) for rank in range(8)]
    full = assemble_frame(blocks, H * W)
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(_bits(full[k][order[1]]), _bits(whole[k])), k
    for world in (2, 8):
        blocks = [render_shard(fn, o, d, rank, world) for rank in range(world)]       # the W ranks, one after the other
        assert len({b.shape for b in blocks}) == 1 and blocks[0].shape[1] == 5        # equal shards (all-gather requirement)
        full = assemble_frame(blocks, H * W)
        for k in ("image", "depth", "weights_sum"):
            assert torch.equal(_bits(full[k]), _bits(whole[k])), (world, k)


def test_frame1080_partitions_on_the_reference_schedule(bonsai):
    from laenerf_amd.dist import assemble_frame, render_shard
    net, r, o, d = bonsai
    fn = _render(r, want_stats=True)
    whole = fn(o, d)
    assert whole["stats"]["rows"] >= H * W and 1 <= whole["stats"]["iterations"] <= 1024
    hit = N(whole["weights_sum"]) > 0
    for world in (2, 8):
        full = assemble_frame([render_shard(fn, o, d, rank, world) for rank in range(world)], H * W)
        assert np.abs(N(full["image"]) - N(whole["image"])).max() <= 1e-5
        assert np.abs(N(full["weights_sum"]) - N(whole["weights_sum"])).max() <= 1e-5
        assert np.abs(N(full["depth"])[hit] - N(whole["depth"])[hit]).max() <= 1e-5
        assert np.array_equal(np.isnan(N(full["depth"])), np.isnan(N(whole["depth"])))
        same = (_bits(full["image"]) == _bits(whole["image"])).all(dim=1).float().mean().item()
        assert same > 0.9                                      # the re-rounding touches few rays


def test_frame1080_subset_against_the_oracle_loop(bonsai, O):
    """512 pixels of the frame as their own call (the reference's loop on N = 512) against the oracle's march / encode / MLP /
    composite driven by the same schedule: schedule and row counts exact, values within the fp16-MLP tolerance"""
    net, r, o, d = bonsai
    net.encoder.embeddings.data = net.encoder.embeddings.data.half().float()
    n = 512
    sel = torch.linspace(0, H * W - 1, n, device=DEV).long()
    os_, ds_ = o[sel].contiguous(), d[sel].contiguous()
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        got = r.render_eval(os_, ds_, bg_color=1, max_steps=1024, T_thresh=1e-4, want_stats=True)
    on, dn, bits = N(os_), N(ds_), N(r.density_bitfield)
    nears, fars = O.near_far_from_aabb(on, dn, [-2, -2, -2, 2, 2, 2], 0.2)
    th = O.to_f16_bits(N(net.encoder.embeddings))
    ws_h, wc_h = O.to_f16_bits(N(net.sigma_net.weights)), O.to_f16_bits(N(net.color_net.weights))
    offs = N(net.encoder.offsets)
    wsum, depth, image = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    alive, rays_t = np.arange(n, dtype=np.int32), nears.copy()
    step, rows, iters = 0, 0, 0
    while step < 1024 and alive.size > 0:
        n_alive = alive.size
        n_step = max(min(n // n_alive, 8), 1)
        xyzs, dirs, deltas = O.march_rays(n_alive, n_step, alive, rays_t, on, dn, 2.0, bits, 2, 128, nears, fars,
                                          np.zeros(n_alive, np.float32))[:3]
        m = n_alive * n_step
        pad = (-m) % 128
        x01 = ((xyzs[:m] + np.float32(2)) * np.float32(0.25)).astype(np.float32)
        enc, _ = O.grid_encode_forward(x01, th, offs, net.encoder.per_level_scale, 16, f16=True, out_blc=True)
        h, _ = O.ffmlp_forward(np.concatenate([enc, np.zeros((pad, 32), np.uint16)]), ws_h, 32, 16, 64, 2)
        hf = O.from_f16_bits(h)[:m]
        sigma = np.exp(hf[:, 0]).astype(np.float32)
        sh, _ = O.sh_encode_forward(dirs[:m], 4)
        cin = O.to_f16_bits(np.concatenate([sh, hf[:, 1:], np.zeros((m, 1), np.float32)], 1))
        oc, _ = O.ffmlp_forward(np.concatenate([cin, np.zeros((pad, 32), np.uint16)]), wc_h, 32, 16, 64, 3)
        rgb = (1 / (1 + np.exp(-O.from_f16_bits(oc)[:m, :3]))).astype(np.float32)
        O.composite_rays(n_alive, n_step, alive, rays_t, sigma, rgb, deltas[:m], wsum, depth, image, 1e-4)
        alive = alive[alive >= 0]
        step += n_step; rows += m; iters += 1
    image = image + (1 - wsum)[:, None]
    assert got["stats"]["iterations"] == iters and got["stats"]["rows"] == rows
    assert (wsum > 0).mean() > 0.2
    hit = wsum > 0
    dref = np.clip(depth - nears, 0, None)[hit] / (fars - nears)[hit]
    dev = {"weights_sum": float(np.abs(N(got["weights_sum"]) - wsum).max()), "image": float(np.abs(N(got["image"]) - image).max()),
           "depth": float(np.abs(N(got["depth"])[hit] - dref).max())}
    print("frame loop vs oracle loop, max abs deviation:", {k: float("%.3g" % v) for k, v in dev.items()})
    for k, v in dev.items():
        assert v < ORACLE_LOOP_TOL[k], dev


def test_frame1080_two_ranks_in_fresh_processes_hold_the_same_frame():
    """`bench.py --workload frame1080` as the driver launches it for N = 2 (torch.distributed.run, one process per rank); on
    this one-GPU box both ranks use cuda:0 and the exchange goes over gloo (LAE_BENCH_DIST_BACKEND / LAE_BENCH_SINGLE_DEVICE).
    Fresh child processes: nothing is exec'ed from this (GPU-initialised) process."""
    env = dict(os.environ, LAE_BENCH_DIST_BACKEND="gloo", LAE_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29500 + (os.getpid() + 1080) % 2000
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "frame1080",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["unit"] == "Mrays/s" and j["value"] > 1
    assert len(j["frame_sha256_per_rank"]) == 2 and j["ranks_hold_the_same_frame"] is True
    assert j["gather_bytes_per_rank"] == 8100 * 128 * 5 * 4
    assert j["config"]["rays_per_frame"] == H * W and j["config"]["rays_hitting_geometry"] > 0.2


def test_default_workload_two_ranks_carries_the_sharded_frame_object():
    """`python bench.py --gpus 2` WITHOUT --workload and WITHOUT torchrun: bench.py starts the two ranks itself
    (a fresh torch.distributed.run child before any GPU call of the parent), relays rank 0's line and exit code.  Beside the
    replica `value` the line must carry the north star's split -- the configs[3] frame ray-sharded over the job's ranks with one
    all-gather per frame, timed, with the speed-up against the BEST one-GPU time of the same frame on rank 0 alone, the backend
    and the world size as the process group reports them (the reference's dormant gather: nerf/utils.py:1555-1570).  Same gloo /
    one-device rehearsal as above (the driver's torchrun form is the previous test)."""
    env = dict(os.environ, LAE_BENCH_DIST_BACKEND="gloo", LAE_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_size"] == 2 and j["backend"] == "gloo"
    assert j["scaling"] == "weak" and j["unit"] == "Mrays/s" and j["value"] > 0
    assert j["windows"]["median"] == j["ms_per_step"]               # `value` is the median window
    f = j["frame1080"]
    assert f["n_gpus"] == 2 and f["world_size"] == 2 and f["backend"] == "gloo"
    assert f["ranks_hold_the_same_frame"] is True and len(f["frame_sha256_per_rank"]) == 2
    assert f["gather_bytes_per_rank"] == 8100 * 128 * 5 * 4 and f["rays"] == H * W
    assert set(f["n1_row_budget_ms"]) == {"reference_rule", "2N", "4N", "8N"}
    assert f["best_n1_ms_per_frame"] == pytest.approx(min(f["n1_row_budget_ms"].values()), rel=1e-3)
    assert f["best_n1_ms_per_frame"] <= f["n1_ms_per_frame"]
    assert f["ms_per_frame"] > 0 and f["speedup_vs_n1"] == pytest.approx(f["best_n1_ms_per_frame"] / f["ms_per_frame"], rel=1e-2)
    assert f["speedup_vs_n1_reference_schedule"] == pytest.approx(f["n1_ms_per_frame"] / f["ms_per_frame"], rel=1e-2)
    assert "eval_frame" not in j                                    # one-GPU extras stay out of the multi-rank line


def test_frame1080_boosted_budget_partition_of_8_matches_the_whole_frame(bonsai):
    """What `bench.py --gpus 8` renders per rank -- a shard of the 1080p bound-2 frame in pixel-tile order with the WHOLE frame's
    row budget (row_budget = 8 x its rays) -- assembled over the 8 ranks, against the whole frame on the reference's rule
    (renderer.py:363) and on the boosted one-GPU schedules bench.py times (2N / 8N rows per iteration): <= 1e-5 everywhere
    (north_star: 1e-4 RGB).  Until round 5 the budget check ran at 6000 rays, bound 1 only (tests/test_gpu_frame.py)."""
    from laenerf_amd.dist import assemble_frame, pixel_tile_order, render_shard
    net, r, o, d = bonsai
    whole = _render(r, want_stats=True)(o, d)
    hit = N(whole["weights_sum"]) > 0
    order = pixel_tile_order((H, W), o.device)
    ot, dt_ = o[order[0]], d[order[0]]
    iters = []

    def shard_fn(ro, rd):
        res = _render(r, want_stats=True, row_budget=8 * ro.shape[0])(ro, rd)
        iters.append(res["stats"]["iterations"])
        return res
    full = assemble_frame([render_shard(shard_fn, ot, dt_, rank, 8) for rank in range(8)], H * W)
    assert max(iters) < whole["stats"]["iterations"] // 2        # the boosted shard really runs a shorter schedule

    def close(a, b):
        assert np.abs(N(a["image"]) - N(b["image"])).max() <= 1e-5
        assert np.abs(N(a["weights_sum"]) - N(b["weights_sum"])).max() <= 1e-5
        assert np.abs(N(a["depth"])[hit] - N(b["depth"])[hit]).max() <= 1e-5
        assert np.array_equal(np.isnan(N(a["depth"])), np.isnan(N(b["depth"])))
    close({k: full[k][order[1]] for k in ("image", "depth", "weights_sum")}, whole)
    for k in (2, 8):
        close(_render(r, row_budget=k * H * W)(o, d), whole)
