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
