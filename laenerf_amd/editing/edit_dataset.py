"""Extraction of the LAENeRF training set from a trained NeRF: `EditDataset.__init__`'s per-view loop
(editing/edit_dataset.py:74-234) on the MI355X kernels -- one full-resolution distill render per training view.

Per view the reference runs `get_rays` (≈15 torch ops over two H*W meshgrids), the host loop of `run_cuda_distill`
(≈80 iterations, one device sync each) and ≈60 masking / cropping torch ops.  Here: `lae_get_rays` (ray generation with
the near bounds in the same pass) -> `lae_render_frame` with the edit grid (device-resident loop) -> the same selection
rules.  The selection itself is plain torch on per-pixel vectors (plumbing, not a hot spot: ≈0.3 ms against a ≈15-45 ms
render).  Results stay on the device unless `to_cpu=True` (the reference stores CPU tensors).
"""
import torch

from ..rays import get_rays

__all__ = ["select_edit_pixels", "extract_view", "extract_views", "min_dist_to_points"]


def select_edit_pixels(weights_density, weights_edit, depth, min_near, depth_diff):
    """edit_dataset.py:91-101: per-pixel weight of the edit region.  A pixel counts when the edit grid explains (almost)
    all of its opacity (|w_density - w_edit| <= depth_diff: rejects floaters in front / behind) and its depth is valid;
    accepted pixels take the density weight.  Returns (pred_w8s [HW], mask indices [K])."""
    w = weights_edit.clone()
    w[torch.abs(weights_density - w) > depth_diff] = 0
    w[depth.reshape(-1) < min_near] = 0
    w[w > 0] = weights_density[w > 0]
    return w, w.nonzero(as_tuple=True)[0]


def min_dist_to_points(pts, points, max_dist):
    """-> (min(max_dist, distance of every pts[i] to its nearest points[j]) [n], their maximum [1]) -- edit_dataset.py:131-143's
    chunked torch.cdist + min + clamp_max as one kernel (lae_min_dist_to_points); fp32 cuda tensors [n,3], [m,3]"""
    from .. import _lib
    pts, points = pts.float().contiguous(), points.float().contiguous()
    _lib.need_cuda(pts, points)
    n, m = pts.shape[0], points.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    d_max = torch.zeros(1, dtype=torch.float32, device=pts.device)       # n == 0: the kernel does not run, the maximum of nothing is 0
    scratch = torch.empty(max(n, 1), dtype=torch.int32, device=pts.device)
    _lib.check(_lib.load().lae_min_dist_to_points(_lib.ptr(pts), n, _lib.ptr(points) if m else None, m, float(max_dist), _lib.ptr(out), _lib.ptr(d_max),
                                                  _lib.ptr(scratch), _lib.stream()), "min_dist_to_points")
    return out, d_max


def _crop_terms(h, w, mask, pred_w8s, target, d_mask, dist_factor=None):
    """edit_dataset.py:194-232: bounding box of the region, ground-truth cut-out, depth TV weights"""
    dev = pred_w8s.device
    m = torch.zeros((h, w), dtype=torch.float32, device=dev)
    m.flatten(0, 1)[mask] = pred_w8s
    x, y = m.nonzero(as_tuple=True)
    x_min, x_max, y_min, y_max = x.min(), x.max(), y.min(), y.max()
    gt = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    gt.flatten(0, 1)[mask] = target.float()
    gt = gt[x_min:x_max, y_min:y_max]
    wts = m[x_min:x_max, y_min:y_max].clone()
    wts[wts < 0.98] = 0
    w_h = wts[:-1, :] * wts[1:, :]
    w_h[1:] *= wts[:-2, :] * wts[2:, :]
    w_v = wts[:, :-1] * wts[:, 1:]
    w_v[:, 1:] *= wts[:, :-2] * wts[:, 2:]
    rgb_h = torch.abs(gt[:-1, :] - gt[1:, :]).sum(-1)
    rgb_v = torch.abs(gt[:, :-1] - gt[:, 1:]).sum(-1)
    depth = torch.zeros((h, w), dtype=torch.float32, device=dev)
    depth.flatten(0, 1)[mask] = d_mask
    depth = depth[x_min:x_max, y_min:y_max]
    out = {"cut_min_max_xy": torch.stack((x_min, x_max, y_min, y_max)), "cut_gt": gt,
           "cut_tv_h": torch.abs(depth[:-1, :] - depth[1:, :]) * w_h * rgb_h,
           "cut_tv_v": torch.abs(depth[:, :-1] - depth[:, 1:]) * w_v * rgb_v}
    if dist_factor is not None:
        wt = torch.zeros((h, w), dtype=torch.float32, device=dev)
        wt.flatten(0, 1)[mask] = dist_factor.float()
        out["cut_smooth_trans"] = wt[x_min:x_max, y_min:y_max]
    return out


@torch.no_grad()
def _pack_view(out, g, rays_d, H, W, image, depth_diff, max_dist, num_steps, to_cpu):
    """selection rules + crop terms of ONE view (edit_dataset.py:89-234) on the slices of a (possibly batched) distill render:
    `out` = render of the edit grid, `g` = render of the grow grid or None; every tensor covers this view's H*W rays"""
    w_density, w_edit, depth, min_near = out["weights"], out["weights_edit"], out["depth"], out["min_near"]
    pred_w8s, mask = select_edit_pixels(w_density, w_edit, depth, min_near, depth_diff)
    if mask.numel() == 0:
        return None
    res = {"weights_densitygrid": w_density, "weights_editgrid": pred_w8s, "pred_imgs": out["image"]}
    dist_factor = None
    if g is not None:                                                         # :122-146 smooth transition weights
        x_grow = g["x_term"][g["weights_edit"] > .99]
        if x_grow.shape[0]:
            min_d, d_max = min_dist_to_points(out["x_term"][mask], x_grow, max_dist)      # :131-143 (cdist in 1000-row chunks there)
            dist_factor = 1 - (min_d / d_max)
        else:
            dist_factor = torch.zeros_like(pred_w8s[mask])
        md = dist_factor.nonzero(as_tuple=True)[0]
        res["indices_interp"], res["dist_weights"] = md, dist_factor[md]
    target = image.to(rays_d.device)
    if target.shape[-1] == 4:
        target = target[..., :3] * target[..., -1][..., None]
    target = target.reshape(-1, 3)[mask]
    d_mask = depth[mask]
    w_sel = pred_w8s[mask]
    res.update(w8s=w_sel, targets=target, x_term=out["x_term"][mask], dirs=rays_d[mask], depths=d_mask, indices=mask,
               depth_factor=(d_mask.max() - d_mask.min()) / num_steps)
    res.update(_crop_terms(H, W, mask, w_sel, target, d_mask, dist_factor))
    if to_cpu:
        res = {k: v.cpu() for k, v in res.items()}
    return res


def _render_views(renderer, poses, intrinsics, H, W, edit_grid, grow_grid):
    """rays of ALL given views in one get_rays launch and ONE device-resident distill render over them (+ one for the grow
    grid): the frame loop takes any number of rays, so V views cost one loop instead of V (the reference renders view by
    view, edit_dataset.py:74-87)"""
    V = poses.shape[0]
    rays = get_rays(poses.reshape(V, 4, 4), intrinsics, H, W, -1, aabb=renderer.aabb_infer, min_near=renderer.min_near)
    rays_o, rays_d, nears = rays["rays_o"].reshape(-1, 3), rays["rays_d"].reshape(-1, 3), rays["nears"].reshape(-1)
    with torch.autocast("cuda", dtype=torch.float16):
        out = renderer.render_distill(rays_o, rays_d, edit_grid, perturb=True, nears=nears)
        g = renderer.render_distill(rays_o, rays_d, grow_grid, perturb=True, grow_grid=True, nears=nears) if grow_grid is not None else None
    n = H * W

    def view_of(d, v):
        if d is None:
            return None
        o = {k: (t[v * n:(v + 1) * n] if torch.is_tensor(t) and t.shape[:1] == (V * n,) else t) for k, t in d.items()}
        o["min_near"] = nears[v * n:(v + 1) * n].min()            # per view, like the reference (renderer.py:478)
        return o
    return [(view_of(out, v), view_of(g, v), rays_d[v * n:(v + 1) * n]) for v in range(V)]


def extract_view(renderer, pose, intrinsics, H, W, edit_grid, image, depth_diff=0.5, grow_grid=None, max_dist=0.1,
                 num_steps=1024, to_cpu=False):
    """One iteration of the loop at edit_dataset.py:74-234.

    renderer: laenerf_amd.renderer.NeRFRenderer (eval mode); pose [4,4] or [1,4,4]; edit_grid / grow_grid: uint8 bitfields;
    image [H,W,3|4] ground truth of the view.  Returns None when the region is occluded in this view (:103-108), else a
    dict with the reference's per-view entries (w8s, targets, x_term, dirs, depths, indices, weights_densitygrid,
    weights_editgrid, pred_imgs, cut_*, depth_factor and, with a grow grid, indices_interp / dist_weights)."""
    (out, g, rays_d), = _render_views(renderer, pose.reshape(1, 4, 4), intrinsics, H, W, edit_grid, grow_grid)
    return _pack_view(out, g, rays_d, H, W, image, depth_diff, max_dist, num_steps, to_cpu)


@torch.no_grad()
def extract_views(renderer, poses, intrinsics, H, W, edit_grid, images, batch_views=4, depth_diff=0.5, grow_grid=None, max_dist=0.1,
                  num_steps=1024, to_cpu=False):
    """the whole loop: returns (list of per-view dicts, list of occluded pose indices) (edit_dataset.py:74, 107).
    `batch_views` views share one ray-generation launch and one distill render (SURVEY 8f-3 "batched multi-view distill
    render"); the selection rules run per view on slices of the batch."""
    views, occluded = [], []
    for i0 in range(0, poses.shape[0], max(1, int(batch_views))):
        batch = poses[i0:i0 + max(1, int(batch_views))]
        for j, (out, g, rays_d) in enumerate(_render_views(renderer, batch, intrinsics, H, W, edit_grid, grow_grid)):
            v = _pack_view(out, g, rays_d, H, W, images[i0 + j], depth_diff, max_dist, num_steps, to_cpu)
            if v is None:
                occluded.append(i0 + j)
            else:
                v["pose_idx"] = i0 + j
                views.append(v)
    return views, occluded
