"""Ray-sharded whole-frame rendering across the GPUs of one node (BASELINE config 4).

Rays are independent units: the frame's N rays are cut into 128-ray tiles dealt round-robin to the ranks
(interleaving balances empty-sky and object tiles), every rank renders its shard with the full inference loop,
and ONE all-gather (RCCL over xGMI; gloo in the CPU tests) of the [n_shard, 5] fp32 block (rgb, depth,
weights_sum) rebuilds the frame on every rank -- the shape of the reference's dormant
`dist.all_gather(preds)` (nerf/utils.py:1560-1562).  No other collective is on the data path.
"""
import torch
import torch.distributed as dist

TILE = 128


def shard_indices(n_rays, rank, world_size, tile=TILE):
    """ray ids owned by `rank`: tiles rank, rank+W, rank+2W, ...; every rank gets the same count (padded with -1)"""
    n_tiles = (n_rays + tile - 1) // tile
    tiles_per_rank = (n_tiles + world_size - 1) // world_size
    t = torch.arange(tiles_per_rank) * world_size + rank
    idx = (t[:, None] * tile + torch.arange(tile)[None, :]).reshape(-1)
    idx[idx >= n_rays] = -1
    return idx


def gather_frame(local_block, n_rays, rank, world_size, tile=TILE, group=None):
    """local_block [n_shard, K] (rows in shard_indices order) -> full [n_rays, K] on every rank"""
    local_block = local_block.contiguous()
    if world_size == 1:
        gathered = [local_block]
    else:
        dev = local_block.device
        if local_block.is_cuda and dist.get_backend(group) == "gloo":      # CPU rehearsal of the RCCL path (tests, 1-GPU boxes)
            local_block = local_block.cpu()
        gathered = [torch.empty_like(local_block) for _ in range(world_size)]
        dist.all_gather(gathered, local_block, group=group)
        gathered = [g.to(dev) for g in gathered]
        local_block = local_block.to(dev)
    out = torch.zeros(n_rays, local_block.shape[1], dtype=local_block.dtype, device=local_block.device)
    for r in range(world_size):
        idx = shard_indices(n_rays, r, world_size, tile).to(local_block.device)
        ok = idx >= 0
        out[idx[ok]] = gathered[r][ok]
    return out


def render_frame_sharded(render_fn, rays_o, rays_d, rank, world_size, group=None):
    """render_fn(rays_o, rays_d) -> dict(image [n,3], depth [n], weights_sum [n]); returns the full frame dict"""
    n = rays_o.shape[0]
    idx = shard_indices(n, rank, world_size).to(rays_o.device)
    safe = idx.clamp(min=0)
    res = render_fn(rays_o[safe], rays_d[safe])
    block = torch.cat([res["image"].float(), res["depth"].float()[:, None], res["weights_sum"].float()[:, None]], dim=1)
    full = gather_frame(block, n, rank, world_size, group=group)
    return {"image": full[:, :3], "depth": full[:, 3], "weights_sum": full[:, 4]}


# ---------------------------------------------------------------------------------------------------------------------
# Data-parallel training.  No configuration of the north star exchanges gradients (training is
# replicas-only there); this is the optional DP mode: every rank marches / shades its own ray batch, the gradients are
# averaged, every rank applies the same Adam step.  The payload is dominated by the hash table's gradient (12.2 M
# parameters = 24.5 MB in the fp16 accumulator FusedAdam owns), so it goes out as ONE flat all-reduce per dtype: on
# 8 MI355X, RCCL runs a ring/tree over the xGMI links (7 x ~153 GB/s per GPU): 2 * 7/8 * 24.5 MB / link rate ~ 0.3 ms,
# i.e. comparable to the 0.5 ms step itself -- which is why the default bench mode keeps independent replicas.
def allreduce_mean_(tensors, world_size=None, group=None, bucket_bytes=64 << 20):
    """in place: every tensor becomes the mean over the ranks.  Tensors are packed by dtype into flat buckets of at most
    `bucket_bytes` (one all-reduce each); fp16 payloads are reduced in fp16 (sum of W values scaled by 1/W first, so the
    reduction cannot overflow where the local gradients did not)."""
    world_size = dist.get_world_size(group) if world_size is None else world_size
    if world_size == 1:
        return tensors
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, dev), ts in by_dtype.items():
        bucket, size = [], 0
        buckets = []
        for t in ts:
            nbytes = t.numel() * t.element_size()
            if bucket and size + nbytes > bucket_bytes:
                buckets.append(bucket); bucket, size = [], 0
            bucket.append(t); size += nbytes
        if bucket:
            buckets.append(bucket)
        for b in buckets:
            flat = torch.cat([t.reshape(-1) for t in b]) if len(b) > 1 else b[0].reshape(-1)
            flat.mul_(1.0 / world_size)
            rehearsal = flat.is_cuda and dist.get_backend(group) == "gloo"        # CPU rehearsal of the RCCL path
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
            grads.append(shadow.grad_half)
        elif p.grad is not None:
            grads.append(p.grad)
    return allreduce_mean_(grads, world_size, group)
