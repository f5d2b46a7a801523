"""Ray-sharded whole-frame rendering across the GPUs of one node (BASELINE config 4).

Rays are independent units: the frame's N rays are cut into 128-ray tiles dealt round-robin to the ranks
(interleaving balances empty-sky and object tiles), every rank renders its shard with the full inference loop,
and ONE all-gather (RCCL over xGMI; gloo in the CPU tests) of the [n_shard, 5] fp32 block (rgb, depth,
weights_sum) rebuilds the frame on every rank -- the shape of the reference's dormant
`dist.all_gather(preds)` (nerf/utils.py:1560-1562).  No other collective is on the data path.
"""
import os

import torch
import torch.distributed as dist

TILE = 128
RS_AG_MIN_ELEMS = 1 << 20      # gradients at least this large go out as reduce-scatter + all-gather (RCCL only)
# World size 1 needs no exchange and every function below returns early for it -- unless this switch is on: then a
# one-rank group still goes through EVERY collective call (all_gather_into_tensor, broadcast, reduce_scatter_tensor,
# all_reduce).  That is how a one-GPU box executes the RCCL entry points, dtypes and stream hand-offs the 8-GPU run uses
# (tests/test_gpu_rccl.py, tools/rccl_w1_check.py); W = 1 moves no data between devices.
FORCE_COLLECTIVES = os.environ.get("LAE_DIST_FORCE_COLLECTIVES") == "1"


def _exchange(world_size):
    return world_size > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def shard_indices(n_rays, rank, world_size, tile=TILE):
    """ray ids owned by `rank`: tiles rank, rank+W, rank+2W, ...; every rank gets the same count (padded with -1)"""
    n_tiles = (n_rays + tile - 1) // tile
    tiles_per_rank = (n_tiles + world_size - 1) // world_size
    t = torch.arange(tiles_per_rank) * world_size + rank
    idx = (t[:, None] * tile + torch.arange(tile)[None, :]).reshape(-1)
    idx[idx >= n_rays] = -1
    return idx


_shard_cache = {}


def shard_indices_device(n_rays, rank, world_size, device, tile=TILE):
    """`shard_indices` clamped to valid ray ids, resident on `device`, computed once per (frame size, rank, world)"""
    key = (n_rays, rank, world_size, tile, str(device))
    idx = _shard_cache.get(key)
    if idx is None:
        if len(_shard_cache) > 64:
            _shard_cache.clear()
        idx = shard_indices(n_rays, rank, world_size, tile).clamp(min=0).to(device)
        _shard_cache[key] = idx
    return idx


def deinterleave(gathered, n_rays, world_size, tile=TILE):
    """[W, n_shard, K] (rank-major, as all_gather_into_tensor lays it out) -> [n_rays, K] in ray order.  Rank r holds tiles
    r, r + W, r + 2W, ...: as a [W, tiles_per_rank, tile, K] array the frame is the transpose of the first two axes -- a
    pure permute + reshape (one strided copy), no index tensors; the padding rows (ray id >= n_rays) are the tail."""
    W, n_shard, K = gathered.shape
    assert W == world_size and n_shard % tile == 0
    return gathered.view(W, n_shard // tile, tile, K).permute(1, 0, 2, 3).reshape(-1, K)[:n_rays]


def gather_frame(local_block, n_rays, rank, world_size, tile=TILE, group=None):
    """local_block [n_shard, K] (rows in shard_indices order) -> full [n_rays, K] on every rank: ONE all_gather_into_tensor
    (RCCL over xGMI: every rank ships its block to its 7 peers over 7 links) + the de-interleave."""
    return deinterleave(exchange_blocks(local_block, world_size, group), n_rays, world_size, tile)


def render_shard(render_fn, rays_o, rays_d, rank, world_size):
    """rank `rank`'s part of the frame: its tiles' rays through render_fn(rays_o, rays_d) -> dict(image [n,3], depth [n],
    weights_sum [n]), packed as the [n_shard, 5] fp32 block (rgb, depth, weights_sum) the all-gather ships"""
    n = rays_o.shape[0]
    idx = shard_indices_device(n, rank, world_size, rays_o.device)          # cached on the device: no host work per frame
    res = render_fn(rays_o[idx], rays_d[idx])
    return torch.cat([res["image"].float(), res["depth"].float()[:, None], res["weights_sum"].float()[:, None]], dim=1)


def assemble_frame(blocks, n_rays, tile=TILE):
    """the W ranks' blocks (rank order) -> frame dict; what gather_frame does after the all-gather, for callers that hold
    every block already (one process rendering the W shards one after the other: tests, frames too large for one call)"""
    full = deinterleave(torch.stack(list(blocks)), n_rays, len(blocks), tile)
    return {"image": full[:, :3], "depth": full[:, 3], "weights_sum": full[:, 4]}


PIXEL_TILE = (8, 8)           # pixels per 2-D tile of the ray order below: TILE = 128 rays = two such tiles
_tile_cache = {}


def pixel_tile_order(image_hw, device, tile_hw=PIXEL_TILE):
    """ray order that walks an H x W image in th x tw pixel tiles (rows of tiles, scanline inside a tile) and its inverse, cached
    on the device; None when the image does not divide into such tiles.  Rays of neighbouring PIXELS march through neighbouring
    cells: a wave of the encoder touches fewer cache lines than on a scanline segment (1080p frame 57.05 -> 54.6 ms, one rank's
    shard of 8: 8.50 -> 8.27 ms) -- and a shard's 128-ray tiles become two 8 x 8 pixel blocks instead of a 128-pixel line."""
    H, W = image_hw
    th, tw = tile_hw
    if H % th or W % tw:
        return None
    key = (H, W, th, tw, str(device))
    if key not in _tile_cache:
        if len(_tile_cache) > 16:
            _tile_cache.clear()
        idx = torch.arange(H * W, device=device).view(H // th, th, W // tw, tw).permute(0, 2, 1, 3).reshape(-1)
        inv = torch.empty_like(idx)
        inv[idx] = torch.arange(H * W, device=device)
        _tile_cache[key] = (idx, inv)
    return _tile_cache[key]


_plan_cache = {}


def frame_plan(n_rays, rank, world_size, device, image_hw=None, tile=TILE):
    """the two index vectors of a sharded frame, composed once per (frame size, rank, world) and kept on the device:
    `take` [n_shard]: the caller's ray ids this rank renders (its 128-ray units of the pixel-tile order when image_hw = (H, W)
    divides into 8 x 8 tiles, of the caller's order otherwise; rotated round-robin, see below; padding rows repeat ray 0);
    `put` [n_rays]: row of the rank-major all-gather buffer that holds the caller's ray q -- the de-interleave of the tiles and the
    way back from pixel-tile order as ONE gather.  Round 6: before, every rank gathered the WHOLE frame's rays into tile order, then
    its shard from that, and undid the two permutations with a strided copy and a second full-frame gather."""
    hw = tuple(image_hw) if image_hw is not None and image_hw[0] * image_hw[1] == n_rays else None
    key = (n_rays, rank, world_size, tile, str(device), hw)
    plan = _plan_cache.get(key)
    if plan is None:
        if len(_plan_cache) > 64:
            _plan_cache.clear()
        order = pixel_tile_order(hw, device) if hw is not None else None
        n_tiles = (n_rays + tile - 1) // tile
        per_rank = (n_tiles + world_size - 1) // world_size
        n_shard = per_rank * tile
        j = torch.arange(n_shard, device=device)
        # the deal: 128-ray unit u = q * W + r goes to rank (r + q) mod W -- round-robin ROTATED by one rank per group of W units.
        # The plain deal (rank r takes units r, r + W, ...: shard_indices) hands a rank the same pixel columns of every tile row
        # whenever the units per tile row divide by W (1080p: 120 per row, W = 8): vertical stripes, and the eight shards of the
        # bench frame differed by 5 % in samples (7.80 ... 8.11 ms); rotated, every rank sees every column
        q_own = j // tile
        pos = (q_own * world_size + (rank - q_own) % world_size) * tile + j % tile      # position in the (tile-ordered) frame of this rank's row j
        pos_c = pos.clamp(max=n_rays - 1)
        take = order[0][pos_c] if order is not None else pos_c
        take = torch.where(pos < n_rays, take, torch.zeros_like(take))
        # all ranks' rows: g = r * n_shard + j  <->  frame position ((j // tile) * W + r) * tile + j % tile
        g = torch.arange(world_size * n_shard, device=device)
        r_, j_ = g // n_shard, g % n_shard
        q_all = j_ // tile
        p_all = (q_all * world_size + (r_ - q_all) % world_size) * tile + j_ % tile
        valid = p_all < n_rays
        put = torch.empty(n_rays, dtype=torch.long, device=device)
        q = order[0][p_all[valid]] if order is not None else p_all[valid]
        put[q] = g[valid]
        plan = _plan_cache[key] = {"take": take, "put": put, "n_shard": n_shard}
    return plan


def render_frame_sharded(render_fn, rays_o, rays_d, rank, world_size, group=None, image_hw=None):
    """render_fn(rays_o, rays_d) -> dict(image [n,3], depth [n], weights_sum [n]); returns the full frame dict in the caller's ray
    order.  image_hw = (H, W): the rays are the pixels of one image in scanline order -- they are dealt to the ranks (and rendered)
    in 2-D pixel-tile order (pixel_tile_order); per-ray results do not depend on the order.  Per frame and rank: one gather of the
    rank's own rays, the render, one `cat` into the [n_shard, 5] block, ONE all_gather_into_tensor, one gather into the caller's
    order (frame_plan)."""
    n = rays_o.shape[0]
    plan = frame_plan(n, rank, world_size, rays_o.device, image_hw)
    take = plan["take"]
    res = render_fn(rays_o.reshape(-1, 3)[take], rays_d.reshape(-1, 3)[take])
    block = torch.cat([res["image"].float(), res["depth"].float()[:, None], res["weights_sum"].float()[:, None]], dim=1)
    full = exchange_blocks(block, world_size, group).reshape(-1, block.shape[1])[plan["put"]]
    return {"image": full[:, :3], "depth": full[:, 3], "weights_sum": full[:, 4]}


def exchange_blocks(local_block, world_size, group=None):
    """[n_shard, K] on every rank -> [W, n_shard, K] rank-major on every rank: the ONE collective of a sharded frame"""
    local_block = local_block.contiguous()
    if not _exchange(world_size):
        return local_block.unsqueeze(0)
    dev = local_block.device
    rehearsal = local_block.is_cuda and dist.get_backend(group) == "gloo"    # CPU rehearsal of the RCCL path (tests, 1-GPU boxes)
    src = local_block.cpu() if rehearsal else local_block
    out = torch.empty((world_size * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)   # rank-major
    dist.all_gather_into_tensor(out, src, group=group)
    out = out.view((world_size,) + tuple(src.shape))
    return out.to(dev) if rehearsal else out


@torch.no_grad()
def broadcast_model_state(module, src=0, group=None):
    """SURVEY.md 8e "replicate model state ... by one ncclBroadcast at load": every parameter and buffer of `module`
    (hash table, MLP weights, density grid / bitfield, aabb) goes out from rank `src` as ONE flat broadcast per dtype.
    The copies go into the parameters THEMSELVES (under no_grad), not into `.data`: `.data` is a separate tensor object with
    its own version counter, so a write through it leaves `p._version` where it was and the fp16 shadow tables a FusedAdam
    keeps (TableShadow.table_half compares versions) would go on serving the pre-broadcast table.  Shadows
    found on the module tree are re-derived here as well, so the result does not hinge on the lazy check."""
    if not dist.is_available() or not dist.is_initialized() or not _exchange(dist.get_world_size(group)):
        return module
    tensors = list(module.parameters()) + [b for b in module.buffers()]
    attrs = [getattr(module, n) for n in ("density_bitfield", "density_grid") if isinstance(getattr(module, n, None), torch.Tensor)]
    seen, uniq = set(), []
    for t in tensors + attrs:
        if t.data_ptr() not in seen and t.numel():
            seen.add(t.data_ptr()); uniq.append(t)
    by_dtype = {}
    for t in uniq:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, dev), ts in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        rehearsal = flat.is_cuda and dist.get_backend(group) == "gloo"
        buf = flat.cpu() if rehearsal else flat
        dist.broadcast(buf, src=src, group=group)
        if rehearsal:
            flat = buf.to(dev)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()      # in place on the parameter: bumps its version
    for m in module.modules():                                                   # fp16 shadows follow the new fp32 values now
        sh = getattr(m, "shadow", None)
        if sh is not None and hasattr(sh, "half"):
            p = getattr(m, "embeddings", None) if hasattr(m, "embeddings") else getattr(m, "weights", None)
            if p is not None:
                sh.half.copy_(p.detach()); sh.version = p._version
    return module


# ---------------------------------------------------------------------------------------------------------------------
# Data-parallel training.  No configuration of the north star exchanges gradients (training is
# replicas-only there); this is the optional DP mode: every rank marches / shades its own ray batch, the gradients are
# averaged, every rank applies the same Adam step.  The payload is dominated by the hash table's gradient (12.2 M
# parameters = 24.5 MB in the fp16 accumulator FusedAdam owns), so it goes out as ONE flat all-reduce per dtype: on
# 8 MI355X, RCCL runs a ring/tree over the xGMI links (7 x ~153 GB/s per GPU): 2 * 7/8 * 24.5 MB / link rate ~ 0.3 ms,
# i.e. comparable to the 0.5 ms step itself -- which is why the default bench mode keeps independent replicas.
# RCCL status: every collective of this file has executed on RCCL with a one-rank group on the one-GPU box
# (tests/test_gpu_rccl.py: all_gather_into_tensor, broadcast, reduce_scatter_tensor + all_gather_into_tensor on fp16,
# all_reduce on fp32); no multi-GPU node was available to any round, the CPU tests rehearse W = 2 over gloo (whose
# all-reduce stands in for reduce-scatter + all-gather).
def allreduce_mean_(tensors, world_size=None, group=None, bucket_bytes=64 << 20):
    """in place: every tensor becomes the mean over the ranks.  Tensors are packed by dtype into flat buckets of at most
    `bucket_bytes` (one all-reduce each); fp16 payloads are reduced in fp16 (sum of W values scaled by 1/W first, so the
    reduction cannot overflow where the local gradients did not)."""
    world_size = dist.get_world_size(group) if world_size is None else world_size
    if not _exchange(world_size):
        return tensors
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, dev), ts in by_dtype.items():
        bucket, size = [], 0
        buckets = []
        for t in ts:
            nbytes = t.numel() * t.element_size()
            if t.numel() >= RS_AG_MIN_ELEMS and t.is_contiguous():
                buckets.append([t])                            # a table gradient travels alone, in place: never concatenated
                continue
            if bucket and size + nbytes > bucket_bytes:
                buckets.append(bucket); bucket, size = [], 0
            bucket.append(t); size += nbytes
        if bucket:
            buckets.append(bucket)
        for b in buckets:
            flat = torch.cat([t.reshape(-1) for t in b]) if len(b) > 1 else b[0].reshape(-1)
            flat.mul_(1.0 / world_size)
            backend = dist.get_backend(group)
            rehearsal = flat.is_cuda and backend == "gloo"        # CPU rehearsal of the RCCL path
            if backend == "nccl" and flat.numel() >= RS_AG_MIN_ELEMS:
                # the table gradient: reduce-scatter, then all-gather -- every rank owns 1/W of the sum in
                # between (where a sharded optimizer step would sit); each phase moves (W-1)/W of the payload over the 7
                # xGMI links at once.  Padded to a multiple of W; smaller payloads stay one all-reduce (latency bound).
                # (FusedAdam's accumulators are flat stores whose length divides by every W <= 8 -- allreduce_gradients hands
                # the whole store over -- so `pad` is 0 on that path and nothing is copied)
                pad = (-flat.numel()) % world_size
                work = torch.cat([flat, flat.new_zeros(pad)]) if pad else flat
                shard = torch.empty(work.numel() // world_size, dtype=work.dtype, device=work.device)
                dist.reduce_scatter_tensor(shard, work, op=dist.ReduceOp.SUM, group=group)
                dist.all_gather_into_tensor(work, shard, group=group)
                if pad:
                    flat.copy_(work[:flat.numel()])
            else:
                buf = flat.cpu() if rehearsal else flat
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
                if rehearsal:
                    flat.copy_(buf)
            if len(b) > 1:
                off = 0
                for t in b:
                    t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
    return tensors


def allreduce_gradients(optimizer, world_size=None, group=None):
    """average the gradients a `laenerf_amd.optim.FusedAdam` is about to consume (fp16 accumulators of the tables it owns,
    fp32 `.grad` of the rest) over the ranks; call between `loss.backward()` and `optimizer.step()`"""
    grads = []
    for p, _, _, shadow, _ in optimizer.items:
        if shadow is not None:
            if p.grad is not None:                           # folded into the accumulator, like FusedAdam._grad does
                shadow.grad_half.add_(p.grad.to(torch.half)); p.grad = None
            if hasattr(shadow, "unreported"):
                shadow.unreported = True                     # the reduced sum may overflow: the optimizer scans the table again
                shadow.mark_all_touched()                    # entries other ranks touched arrive through the reduction
            # the accumulator's backing store, zero-padded to a multiple of 840 elements (every W <= 8 divides it): the
            # reduce-scatter needs equal shards and must not copy 24.5 MB per step to get them
            grads.append(getattr(shadow, "grad_store", shadow.grad_half))
        elif p.grad is not None:
            grads.append(p.grad)
    return allreduce_mean_(grads, world_size, group)
