"""world_size-2 gloo test of the ray-shard / all-gather path (no GPU)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_rays, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laenerf_amd.dist import render_frame_sharded
    g = torch.Generator().manual_seed(0)
    o, d = torch.randn(n_rays, 3, generator=g), torch.randn(n_rays, 3, generator=g)

    def fake_render(ro, rd):           # any per-ray function: the frame must come back identical on every rank
        return {"image": torch.stack([ro[:, 0] + rd[:, 1], ro[:, 1] * rd[:, 2], ro[:, 2] - rd[:, 0]], 1),
                "depth": (ro * rd).sum(-1), "weights_sum": ro.norm(dim=-1)}
    full = render_frame_sharded(fake_render, o, d, rank, world)
    ref = fake_render(o, d)
    ok = all(torch.allclose(full[k], ref[k]) for k in ref)
    if n_rays == 128 * 6:                  # the same frame as a 24 x 32 image, dealt in 8 x 8 pixel tiles: same rays back in the caller's order
        tiled = render_frame_sharded(fake_render, o, d, rank, world, image_hw=(24, 32))
        ok = ok and all(torch.equal(tiled[k], full[k]) for k in ref)
        odd = render_frame_sharded(fake_render, o, d, rank, world, image_hw=(12, 64))   # 12 rows do not divide into 8-row tiles: scanline order
        ok = ok and all(torch.equal(odd[k], full[k]) for k in ref)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rays", [1000, 128 * 6, 77])
def test_sharded_frame_gloo(n_rays):
    world = 2
    port = 29500 + (os.getpid() + n_rays) % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def test_shard_indices_partition():
    from laenerf_amd.dist import shard_indices
    for n, w in ((1000, 2), (2073600, 8), (77, 4), (128, 8)):
        all_idx = torch.cat([shard_indices(n, r, w) for r in range(w)])
        assert len({shard_indices(n, r, w).numel() for r in range(w)}) == 1      # equal shard sizes (all-gather)
        real = all_idx[all_idx >= 0]
        assert real.numel() == n and torch.equal(torch.sort(real).values, torch.arange(n))


def test_deinterleave_is_a_pure_reshape():
    """gather_frame's de-interleave of the rank-major all-gather buffer == scattering every rank's rows by shard_indices"""
    from laenerf_amd.dist import deinterleave, shard_indices
    for n, w in ((1000, 2), (128 * 6, 2), (77, 4), (2073600, 8), (5000, 3)):
        blocks = []
        for r in range(w):
            idx = shard_indices(n, r, w)
            blk = torch.stack([idx.float(), idx.float() * 2 + 1], 1)            # row content = its ray id (padding: -1)
            blocks.append(blk)
        full = deinterleave(torch.stack(blocks), n, w)
        assert full.shape == (n, 2)
        assert torch.equal(full[:, 0], torch.arange(n).float()) and torch.equal(full[:, 1], torch.arange(n).float() * 2 + 1)


def _bcast_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laenerf_amd.dist import broadcast_model_state
    torch.manual_seed(100 + rank)                                # different weights on every rank before the broadcast
    m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    m.register_buffer("bits", torch.randint(0, 255, (64,), dtype=torch.uint8))
    m.density_bitfield = torch.randint(0, 255, (32,), dtype=torch.uint8)
    versions = [p._version for p in m.parameters()]
    broadcast_model_state(m, src=0)
    # the copies land in the parameters themselves: their version counters move, which is what the fp16 shadow tables of a
    # FusedAdam watch (a copy through `.data` leaves `_version` alone -- ADVICE r2)
    assert all(p._version > v for p, v in zip(m.parameters(), versions))
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref_bits = torch.randint(0, 255, (64,), dtype=torch.uint8)
    ref_field = torch.randint(0, 255, (32,), dtype=torch.uint8)
    ok = all(torch.equal(a, b) for a, b in zip(m.parameters(), ref.parameters()))
    ok &= bool(torch.equal(m.bits, ref_bits)) and bool(torch.equal(m.density_bitfield, ref_field))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_broadcast_model_state_gloo():
    world = 2
    port = 29500 + (os.getpid() + 333) % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laenerf_amd.dist import allreduce_mean_
    g = torch.Generator().manual_seed(100 + rank)
    ts = [torch.randn(1000, 2, generator=g).half(), torch.randn(7168, generator=g).half(), torch.randn(11264, generator=g).half(),
          torch.randn(8, 3, generator=g), torch.randn(5, generator=g)]
    every = [[torch.randn(1000, 2, generator=torch.Generator().manual_seed(100 + r)).half()] for r in range(world)]
    mine = [t.clone() for t in ts]
    allreduce_mean_(ts, bucket_bytes=20000)                   # small buckets: the fp16 tensors split into two all-reduces
    ok = True
    # reference: mean over ranks of the first tensor, computed locally from the same seeds
    ref0 = sum(e[0].float() / world for e in every)
    ok &= bool(torch.allclose(ts[0].float(), ref0, atol=2e-3))
    ok &= all(t.dtype == m.dtype and t.shape == m.shape for t, m in zip(ts, mine))
    # every rank must end with identical tensors: compare through a second all-reduce of the difference to rank 0's copy
    for t in ts:
        ref = t.clone(); dist.broadcast(ref, src=0)
        ok &= bool(torch.equal(ref, t))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_dp_gradient_allreduce_gloo():
    world = 2
    port = 29500 + (os.getpid() + 777) % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def _forced_w1_worker(port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LAE_DIST_FORCE_COLLECTIVES="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from laenerf_amd import dist as D
    assert D.FORCE_COLLECTIVES
    seen = []
    for name in ("all_gather_into_tensor", "all_reduce", "broadcast"):
        fn = getattr(dist, name)
        setattr(dist, name, (lambda f, n: (lambda *a, **k: (seen.append(n), f(*a, **k))[1]))(fn, name))
    n = 1000
    idx = D.shard_indices(n, 0, 1).clamp(min=0)
    full = D.gather_frame(torch.stack([idx.float(), idx.float() * 3], 1), n, 0, 1)
    ok = bool(torch.equal(full[:, 0], torch.arange(n).float()) and torch.equal(full[:, 1], torch.arange(n).float() * 3))
    ts = [torch.randn(4000).half(), torch.randn(33)]
    ref = [t.clone() for t in ts]
    D.allreduce_mean_(ts, 1)
    ok &= all(torch.equal(a, b) for a, b in zip(ts, ref))            # the mean over one rank is the identity, bit for bit
    m = torch.nn.Linear(3, 2)
    w = m.weight.detach().clone()
    D.broadcast_model_state(m, src=0)
    ok &= bool(torch.equal(m.weight, w))
    ret[0] = ok and set(seen) == {"all_gather_into_tensor", "all_reduce", "broadcast"}
    dist.destroy_process_group()


def test_forced_collectives_with_one_rank():
    """LAE_DIST_FORCE_COLLECTIVES=1: a one-rank group still goes through every collective call (how the one-GPU box executes
    the RCCL entry points, tests/test_gpu_rccl.py); here over gloo"""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_forced_w1_worker, args=(29500 + (os.getpid() + 999) % 2000, ret))
    p.start(); p.join(120)
    assert p.exitcode == 0 and ret[0]


def test_frame_plan_composes_the_tile_deal_and_the_pixel_tile_order():
    """round 6: dist.frame_plan's two index vectors (`take`: the caller's ray ids a rank renders; `put`: for every ray its row in the
    rank-major all-gather buffer): the ranks' takes partition the frame (every ray exactly once, equal shard sizes), unit u = q W + r
    of the pixel-tile order goes to rank (r + q) mod W (rotated round-robin: no rank keeps the same pixel columns in every tile row),
    and `put` brings the gathered rows back into the caller's order -- for frames that divide into 8 x 8 tiles, frames that do not,
    ragged unit counts and W = 1 ... 8"""
    import torch
    from laenerf_amd import dist as D
    cpu = torch.device("cpu")
    for (H, W) in ((24, 32), (12, 64), (40, 56), (7, 19), (64, 128)):
        n = H * W
        vals = torch.arange(n).float()
        order = D.pixel_tile_order((H, W), cpu)
        tiled = vals[order[0]] if order is not None else vals              # ray id at every position of the dealt order
        for world in (1, 2, 3, 8):
            blocks, seen = [], []
            for r in range(world):
                plan = D.frame_plan(n, r, world, cpu, (H, W))
                assert plan["take"].shape[0] == plan["n_shard"] and plan["n_shard"] % D.TILE == 0
                blocks.append(vals[plan["take"]])
                n_units = -(-n // D.TILE)
                for q in range(plan["n_shard"] // D.TILE):                  # this rank's q-th unit is unit q W + (r - q) mod W
                    u = q * world + (r - q) % world
                    lo, hi = u * D.TILE, min((u + 1) * D.TILE, n)
                    if u < n_units:
                        assert torch.equal(blocks[-1][q * D.TILE:q * D.TILE + hi - lo], tiled[lo:hi]), (H, W, world, r, q)
                        seen.append(torch.arange(lo, hi))
            assert len({b.shape[0] for b in blocks}) == 1                              # equal shards: the all-gather's requirement
            assert torch.equal(torch.sort(torch.cat(seen)).values, torch.arange(n))   # a partition of the frame
            full = torch.stack(blocks).reshape(-1)[plan["put"]]
            assert torch.equal(full, vals), (H, W, world)
    # W = 1 is the plain order; without image_hw (or with one that does not match the ray count) the deal is over the caller's order
    assert torch.equal(D.frame_plan(1000, 0, 1, cpu, None)["take"][:1000], torch.arange(1000))
    plan = D.frame_plan(1000, 1, 4, cpu, None)
    assert torch.equal(plan["take"][:128], torch.arange(128, 256)) and torch.equal(plan["take"][128:256], torch.arange(512, 640))   # q = 1: unit 4 + (1 - 1) % 4
