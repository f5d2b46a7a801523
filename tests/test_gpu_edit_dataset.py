"""-m gpu: the LAENeRF training-set extraction (editing/edit_dataset.py:74-234) on get_rays + the device-resident
distill render: selection rules and crop terms against a numpy restatement of the reference's lines, end to end on a
synthetic scene, occluded views, grow-grid transition weights."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, N, T
from test_gpu_frame import make

pytestmark = pytest.mark.gpu


def poses_looking_at_origin(n, radius, seed):
    rng = np.random.default_rng(seed)
    P = np.zeros((n, 4, 4), np.float32)
    for i in range(n):
        p = rng.standard_normal(3); p = p / np.linalg.norm(p) * radius
        f = -p / np.linalg.norm(p)
        up = np.array([0, 0, 1.0]) if abs(f[2]) < 0.9 else np.array([0, 1.0, 0])
        r = np.cross(up, f); r /= np.linalg.norm(r)
        u = np.cross(f, r)
        P[i, :3, 0], P[i, :3, 1], P[i, :3, 2], P[i, :3, 3], P[i, 3, 3] = r, u, f, p, 1
    return P


def np_select(wd, we, depth, min_near, depth_diff):
    """edit_dataset.py:91-101 in numpy"""
    w = we.copy()
    w[np.abs(wd - w) > depth_diff] = 0
    w[depth < min_near] = 0
    w[w > 0] = wd[w > 0]
    return w, np.nonzero(w)[0]


def test_selection_rules_and_crop_terms():
    from laenerf_amd.editing.edit_dataset import _crop_terms, select_edit_pixels
    rng = np.random.default_rng(0)
    h, w = 40, 56
    wd = rng.random(h * w).astype(np.float32)
    we = np.where(rng.random(h * w) < 0.5, wd * rng.random(h * w), 0).astype(np.float32)
    we[(np.arange(h * w) // w > 30) | (np.arange(h * w) % w < 5)] = 0                 # region away from two borders
    we[rng.random(h * w) < 0.2] = wd[rng.random(h * w) < 0.2].mean() * 0 + 1.0       # saturated pixels (>= 0.98 weights)
    wd = np.maximum(wd, we)
    depth = (rng.random(h * w) * 3).astype(np.float32)
    sel, mask = select_edit_pixels(T(wd), T(we), T(depth), torch.tensor(0.4, device=DEV), 0.5)
    sel0, mask0 = np_select(wd, we, depth, 0.4, 0.5)
    assert np.array_equal(N(sel), sel0) and np.array_equal(N(mask), mask0) and 0 < mask0.size < h * w
    target = rng.random((mask0.size, 3)).astype(np.float32)
    out = _crop_terms(h, w, mask, sel[mask], T(target), T(depth)[mask])
    m = np.zeros(h * w, np.float32); m[mask0] = sel0[mask0]; m = m.reshape(h, w)     # :194-232 in numpy
    x, y = np.nonzero(m)
    x0, x1, y0, y1 = x.min(), x.max(), y.min(), y.max()
    assert np.array_equal(N(out["cut_min_max_xy"]), [x0, x1, y0, y1])
    gt = np.zeros((h * w, 3), np.float32); gt[mask0] = target; gt = gt.reshape(h, w, 3)[x0:x1, y0:y1]
    wt = m[x0:x1, y0:y1].copy(); wt[wt < 0.98] = 0
    wh = wt[:-1] * wt[1:]; wh[1:] *= wt[:-2] * wt[2:]
    wv = wt[:, :-1] * wt[:, 1:]; wv[:, 1:] *= wt[:, :-2] * wt[:, 2:]
    dp = np.zeros(h * w, np.float32); dp[mask0] = depth[mask0]; dp = dp.reshape(h, w)[x0:x1, y0:y1]
    tv_h = np.abs(dp[:-1] - dp[1:]) * wh * np.abs(gt[:-1] - gt[1:]).sum(-1)
    tv_v = np.abs(dp[:, :-1] - dp[:, 1:]) * wv * np.abs(gt[:, :-1] - gt[:, 1:]).sum(-1)
    assert np.array_equal(N(out["cut_gt"]), gt)
    assert np.allclose(N(out["cut_tv_h"]), tv_h, rtol=1e-6, atol=1e-7) and np.allclose(N(out["cut_tv_v"]), tv_v, rtol=1e-6, atol=1e-7)
    assert tv_h.max() > 0


@pytest.mark.parametrize("bound", [1, 2])
def test_extract_views_end_to_end(O, bound):
    """bound 2 = the shape of the config this path serves (configs[4], mip360/bonsai: scripts/configs_mip360/bonsai.sh:2 -> 2
    cascades, cameras inside the box); bound 1 = the lego shape"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.editing import extract_view, extract_views
    from laenerf_amd.rays import get_rays
    net, r = make(bound=bound, seed=2)
    H = W = 96
    intr = np.array([133.3, 133.3, 48.0, 48.0], np.float32)
    poses = T(poses_looking_at_origin(3, 3.2 if bound == 1 else 1.7, seed=1))
    r.density_scale = 30.0                                                  # opaque surfaces: weights saturate (> .99)
    dens = S.sphere_density_grid(cascade=r.cascade, bound=float(bound))     # [C, 128^3]
    coords = O.morton3D_invert(np.arange(128 ** 3, dtype=np.int32))       # cell x coordinate of every Morton index
    right = (coords[:, 0] >= 64)
    edit = T(S.pack_bits_np(np.where(right[None], dens, 0), 10.0))          # edit region: the x > 0 half of the geometry (every cascade)
    grow = T(S.pack_bits_np(np.where(~right[None], dens, 0), 10.0))         # "grow" region: the other half
    images = torch.rand(3, H, W, 4, device=DEV)
    torch.manual_seed(1)
    v = extract_view(r, poses[0], intr, H, W, edit, images[0], depth_diff=0.5, grow_grid=grow)
    assert v is not None
    # the same render, operator by operator, and the reference's lines in numpy
    rays = get_rays(poses[:1], intr, H, W, -1)
    o, d = rays["rays_o"].view(-1, 3), rays["rays_d"].view(-1, 3)
    torch.manual_seed(1)
    with torch.autocast("cuda", dtype=torch.float16):
        out = r.render_distill(o, d, edit, perturb=True)
    wd, we, depth = N(out["weights"]), N(out["weights_edit"]), N(out["depth"])
    sel0, mask0 = np_select(wd, we, depth, float(out["min_near"]), 0.5)
    assert 50 < mask0.size < H * W
    assert np.array_equal(N(v["indices"]), mask0) and np.array_equal(N(v["w8s"]), sel0[mask0])
    assert np.array_equal(N(v["weights_editgrid"]), sel0) and np.array_equal(N(v["weights_densitygrid"]), wd)
    assert np.array_equal(N(v["x_term"]), N(out["x_term"])[mask0]) and np.array_equal(N(v["dirs"]), N(d)[mask0])
    assert np.array_equal(N(v["depths"]), depth[mask0])
    tgt = N(images[0]).reshape(-1, 4)
    assert np.array_equal(N(v["targets"]), (tgt[:, :3] * tgt[:, 3:])[mask0])
    assert np.isclose(float(v["depth_factor"]), (depth[mask0].max() - depth[mask0].min()) / 1024)
    # transition weights: 1 - min distance to the grow-grid surface / max, zero entries dropped (:122-146)
    dw = N(v["dist_weights"])
    assert v["indices_interp"].numel() == dw.size and 0 < dw.size < mask0.size and (dw > 0).all() and dw.max() <= 1.0
    ii = N(v["indices_interp"])                                             # pixels near the seam x = 0 only
    assert np.abs(N(v["x_term"])[ii][:, 0]).max() <= 0.1 + 1e-3
    assert N(v["cut_smooth_trans"]).shape == N(v["cut_gt"]).shape[:2]
    # the whole loop; an edit grid nothing projects to -> every view occluded
    views, occluded = extract_views(r, poses, intr, H, W, edit, images, depth_diff=0.5, to_cpu=True)
    assert len(views) == 3 and occluded == [] and all(not t.is_cuda for t in views[0].values() if torch.is_tensor(t))
    views, occluded = extract_views(r, poses, intr, H, W, torch.zeros_like(edit), images)
    assert views == [] and occluded == [0, 1, 2]


def test_batched_views_equal_single_views_up_to_the_jitter(O):
    """extract_views(batch_views=3): one ray-generation launch and one distill render for three views; per view the same
    entries as the view-by-view loop (the jittered march start differs per call, so pixel sets agree to a few edge pixels)"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.editing import extract_views
    net, r = make(bound=1, seed=2)
    H = W = 96
    intr = np.array([133.3, 133.3, 48.0, 48.0], np.float32)
    poses = T(poses_looking_at_origin(3, 3.2, seed=1))
    r.density_scale = 30.0
    dens = S.sphere_density_grid()
    coords = O.morton3D_invert(np.arange(128 ** 3, dtype=np.int32))
    edit = T(S.pack_bits_np(np.where(coords[:, 0] >= 64, dens[0], 0)[None], 10.0))
    images = torch.rand(3, H, W, 3, device=DEV)
    single, occ1 = extract_views(r, poses, intr, H, W, edit, images, batch_views=1)
    batched, occ3 = extract_views(r, poses, intr, H, W, edit, images, batch_views=3)
    assert occ1 == occ3 == [] and [v["pose_idx"] for v in batched] == [0, 1, 2]
    for a, b in zip(single, batched):
        ia, ib = set(N(a["indices"]).tolist()), set(N(b["indices"]).tolist())
        assert len(ia ^ ib) <= max(3, 0.03 * max(len(ia), len(ib))) and len(ia) > 10
        assert a["weights_densitygrid"].shape == b["weights_densitygrid"].shape == (H * W,)
        assert np.abs(N(a["weights_densitygrid"]) - N(b["weights_densitygrid"])).mean() < 5e-3
        assert set(a.keys()) == set(b.keys())


@pytest.mark.parametrize("n,m", [(5003, 20011), (1, 7), (1024, 1024), (777, 0), (4096, 300000)])
def test_min_dist_to_points_equals_the_chunked_cdist(n, m):
    """lae_min_dist_to_points against the reference's lines (edit_dataset.py:131-143): torch.cdist in 1000-row chunks, min over
    the grow points, clamp_max(max_dist), and the maximum it divides by.  Clustered points: many distances below max_dist."""
    from laenerf_amd.editing.edit_dataset import min_dist_to_points
    g = torch.Generator(device="cpu").manual_seed(n * 7 + m)
    pts = (torch.rand(n, 3, generator=g) * 0.8 - 0.4).to(DEV)
    points = (torch.rand(m, 3, generator=g) * 0.5 - 0.1).to(DEV)
    got, got_max = min_dist_to_points(pts, points, 0.1)
    if m == 0:
        assert torch.equal(got, torch.full((n,), 0.1, device=DEV)) and float(got_max) == pytest.approx(0.1)
        return
    # in float64 (exact to fp32 rounding).  torch.cdist in fp32 on this stack returns zeros for a trailing chunk of a few rows
    # (n = 5003: rows 5000-5002), so the reference's own fp32 call is not the yardstick here
    mins = [torch.cdist(pts[i:i + 500].double(), points.double()).min(dim=-1).values for i in range(0, n, 500)]
    ref = torch.clamp_max(torch.cat(mins), 0.1).float()
    assert got.shape == ref.shape and torch.allclose(got, ref, rtol=1e-6, atol=1e-8)
    assert float(got_max) == float(got.max()) and float(got_max) == pytest.approx(float(ref.max()), rel=1e-6)
    assert (got < 0.1).any() or n < 10
