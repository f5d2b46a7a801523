"""Ray generation of the data loader: `get_rays` with the signature and results of nerf/utils.py:61-153.

The reference builds two H*W meshgrids, gathers the sampled pixels out of them, stacks, normalises and multiplies by the
pose (about 15 torch launches and two H*W temporaries per call, once per training step in `NeRFDataset.collate`,
nerf/provider.py:330-365, and once per view in `EditDataset`, editing/edit_dataset.py:74-77).  Here the pixel sampling
keeps torch's generators (same draws, same order, so a seed selects the same pixels) and everything after it is ONE HIP
kernel (`lae_get_rays`), which can also emit the ray/box interval of `near_far_from_aabb` in the same pass.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream

__all__ = ["get_rays", "custom_meshgrid"]


def custom_meshgrid(*args):
    """nerf/utils.py:43-48"""
    return torch.meshgrid(*args, indexing="ij")


@torch.no_grad()
def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1, perturb_ray_dirs=False, aabb=None, min_near=0.2):
    """poses [B,4,4] cam2world, intrinsics (fx, fy, cx, cy) -> {'rays_o', 'rays_d': [B,N,3], 'inds': [B,N] (N > 0),
    'inds_coarse' (error_map)}; N <= 0: all H*W pixels.  MI355X-native extra: aabb (6 floats) adds 'nears', 'fars' [B,N]
    (raymarching.near_far_from_aabb(rays_o, rays_d, aabb, min_near)) from the same kernel."""
    if not poses.is_cuda:
        raise RuntimeError("laenerf_amd.get_rays: poses must live on the GPU (there is no CPU fallback)")
    device = poses.device
    poses = poses.to(torch.float32).contiguous()
    B = poses.shape[0]
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    results = {}
    inds = None
    if N > 0:
        N = min(N, H * W)
        if patch_size > 1:                                                   # :90-107 (error_map ignored)
            num_patch = N // (patch_size ** 2)
            inds_x = torch.randint(0, H - patch_size, size=[num_patch], device=device)
            inds_y = torch.randint(0, W - patch_size, size=[num_patch], device=device)
            inds = torch.stack([inds_x, inds_y], dim=-1)
            pi, pj = custom_meshgrid(torch.arange(patch_size, device=device), torch.arange(patch_size, device=device))
            offsets = torch.stack([pi.reshape(-1), pj.reshape(-1)], dim=-1)
            inds = (inds.unsqueeze(1) + offsets.unsqueeze(0)).view(-1, 2)
            inds = inds[:, 0] * W + inds[:, 1]
            inds = inds.expand([B, inds.shape[0]])
        elif error_map is None:                                               # :109-111
            inds = torch.randint(0, H * W, size=[N], device=device)
            inds = inds.expand([B, N])
        else:                                                                 # :112-124
            inds_coarse = torch.multinomial(error_map.to(device), N, replacement=False)
            inds_x, inds_y = inds_coarse // 128, inds_coarse % 128
            sx, sy = H / 128, W / 128
            inds_x = (inds_x * sx + torch.rand(B, N, device=device) * sx).long().clamp(max=H - 1)
            inds_y = (inds_y * sy + torch.rand(B, N, device=device) * sy).long().clamp(max=W - 1)
            inds = inds_x * W + inds_y
            results["inds_coarse"] = inds_coarse
        results["inds"] = inds
    off_x = off_y = 0.0
    if perturb_ray_dirs:                                                      # :133 (CPU generator, like the reference)
        offset = torch.rand(2) - 0.5
        off_x, off_y = float(offset[0]), float(offset[1])
    if inds is None:
        n, ip, stride = H * W, None, 0
    else:
        n = inds.shape[-1]
        shared = inds.stride(0) == 0 or B == 1                               # expand()ed row: one index list for all poses
        src = (inds[0] if shared else inds).to(torch.int64).contiguous()
        ip, stride = ptr(src), (0 if shared else n)
    rays_o = torch.empty(B, n, 3, dtype=torch.float32, device=device)
    rays_d = torch.empty(B, n, 3, dtype=torch.float32, device=device)
    ab = nears = fars = None
    if aabb is not None:
        ab = torch.as_tensor(aabb, dtype=torch.float32, device=device).contiguous()
        nears = torch.empty(B, n, dtype=torch.float32, device=device)
        fars = torch.empty(B, n, dtype=torch.float32, device=device)
    check(_lib.load().lae_get_rays(ptr(poses), B, fx, fy, cx, cy, H, W, ip, stride, n, int(bool(perturb_ray_dirs)), off_x, off_y,
                                   ptr(rays_o), ptr(rays_d), ptr(ab) if ab is not None else None, float(min_near),
                                   ptr(nears) if nears is not None else None, ptr(fars) if fars is not None else None,
                                   stream()), "get_rays")
    results["rays_o"] = rays_o
    results["rays_d"] = rays_d
    if ab is not None:
        results["nears"], results["fars"] = nears, fars
    return results
