"""-m gpu: get_rays (nerf/utils.py:61-153) -- the HIP kernel against the oracle (bit for bit: same operations,
correctly rounded division / sqrt on both sides), against the vectors of the reference's own function, and the
Python operator's sampling modes."""
import numpy as np
import pytest
import torch

from conftest import golden
from gpu_util import DEV, N, T

pytestmark = pytest.mark.gpu
CASES = ("all", "rand", "perturb", "patch", "emap", "lego")


def call(g, tag, aabb=None, min_near=0.2):
    from laenerf_amd import _lib
    from laenerf_amd._lib import check, ptr, stream
    H, W, _ = (int(v) for v in g[f"{tag}_cfg"])
    poses = T(g[f"{tag}_poses"])
    B = poses.shape[0]
    fx, fy, cx, cy = (float(v) for v in g[f"{tag}_intr"])
    inds = T(g[f"{tag}_inds"]) if f"{tag}_inds" in g.files else None
    n = H * W if inds is None else inds.shape[-1]
    off = g[f"{tag}_offset"] if f"{tag}_offset" in g.files else None
    ro = torch.empty(B, n, 3, device=DEV); rd = torch.empty(B, n, 3, device=DEV)
    ab = T(np.asarray(aabb, np.float32)) if aabb is not None else None
    nears = torch.empty(B, n, device=DEV) if ab is not None else None
    fars = torch.empty(B, n, device=DEV) if ab is not None else None
    check(_lib.load().lae_get_rays(ptr(poses), B, fx, fy, cx, cy, H, W, ptr(inds), n if inds is not None else 0, n,
                                   0 if off is None else 1, 0.0 if off is None else float(off[0]),
                                   0.0 if off is None else float(off[1]), ptr(ro), ptr(rd), ptr(ab), min_near, ptr(nears),
                                   ptr(fars), stream()), "get_rays")
    return ro, rd, nears, fars


@pytest.mark.parametrize("tag", CASES)
def test_get_rays_kernel_vs_oracle_and_reference(O, tag):
    g = golden("get_rays")
    H, W, _ = (int(v) for v in g[f"{tag}_cfg"])
    inds = g[f"{tag}_inds"] if f"{tag}_inds" in g.files else None
    off = g[f"{tag}_offset"] if f"{tag}_offset" in g.files else None
    ro, rd, _, _ = call(g, tag)
    ro0, rd0 = O.get_rays(g[f"{tag}_poses"], g[f"{tag}_intr"], H, W, inds=inds, offset=off)
    assert np.array_equal(N(ro), ro0) and np.array_equal(N(rd), rd0)                  # bit-exact vs the oracle
    assert np.array_equal(N(ro), g[f"{tag}_rays_o"])
    assert np.abs(N(rd) - g[f"{tag}_rays_d"]).max() < 3e-7                             # vs the reference's torch ops


def test_get_rays_fused_near_far_equals_the_operator(O):
    from laenerf_amd import raymarching as rm
    g = golden("get_rays")
    for tag, aabb in (("lego", [-1, -1, -1, 1, 1, 1]), ("rand", [-2, -2, -2, 2, 2, 2]), ("all", [-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])):
        ro, rd, nears, fars = call(g, tag, aabb=aabb, min_near=0.05)
        n0, f0 = rm.near_far_from_aabb(ro.view(-1, 3), rd.view(-1, 3), T(np.asarray(aabb, np.float32)), 0.05)
        assert np.array_equal(N(nears).reshape(-1), N(n0)) and np.array_equal(N(fars).reshape(-1), N(f0))
        n1, f1 = O.near_far_from_aabb(N(ro).reshape(-1, 3), N(rd).reshape(-1, 3), aabb, 0.05)
        assert np.array_equal(N(nears).reshape(-1), n1) and np.array_equal(N(fars).reshape(-1), f1)


def test_get_rays_operator_modes(O):
    """the Python operator: same keys / shapes / sampling rules as the reference's function (pixel draws come from torch's
    device generator, so they are checked structurally and the rays against the oracle on the returned indices)"""
    from laenerf_amd.rays import get_rays
    g = golden("get_rays")
    poses = T(g["emap_poses"])
    intr = g["emap_intr"]
    H, W = 300, 200
    res = get_rays(poses, intr, H, W, -1)
    assert set(res) == {"rays_o", "rays_d"} and res["rays_d"].shape == (2, H * W, 3)
    ro0, rd0 = O.get_rays(N(poses), intr, H, W)
    assert np.array_equal(N(res["rays_d"]), rd0) and np.array_equal(N(res["rays_o"]), ro0)
    torch.manual_seed(3)
    res = get_rays(poses, intr, H, W, 500)
    inds = N(res["inds"])
    assert inds.shape == (2, 500) and np.array_equal(inds[0], inds[1]) and inds.min() >= 0 and inds.max() < H * W
    assert np.array_equal(N(res["rays_d"]), O.get_rays(N(poses), intr, H, W, inds=inds)[1])
    torch.manual_seed(3)
    assert np.array_equal(N(get_rays(poses, intr, H, W, 500)["inds"]), inds)            # seeded draws repeat
    res = get_rays(poses[:1], intr, H, W, 64, patch_size=4)                             # 4 patches of 4x4 pixels
    pi = N(res["inds"])[0].reshape(4, 16)
    for p in pi:
        r, c = p // W, p % W
        assert np.array_equal(r - r[0], np.repeat(np.arange(4), 4)) and np.array_equal(c - c[0], np.tile(np.arange(4), 4))
    emap = torch.rand(2, 128 * 128, device=DEV) + 0.01
    res = get_rays(poses, intr, H, W, 96, error_map=emap)
    assert set(res) == {"rays_o", "rays_d", "inds", "inds_coarse"}
    inds, coarse = N(res["inds"]), N(res["inds_coarse"])
    assert inds.shape == (2, 96) and not np.array_equal(inds[0], inds[1])
    for pix, c, size in ((inds // W, coarse // 128, H), (inds % W, coarse % 128, W)):   # a pixel inside its coarse cell (:118-121)
        assert (pix >= np.floor(c * (size / 128))).all() and (pix <= np.minimum(np.floor((c + 1) * (size / 128)), size - 1)).all()
    assert np.array_equal(N(res["rays_d"]), O.get_rays(N(poses), intr, H, W, inds=inds)[1])
    torch.manual_seed(5)
    off = (torch.rand(2) - 0.5).numpy()
    torch.manual_seed(5)
    res = get_rays(poses[:1], intr, H, W, -1, perturb_ray_dirs=True)
    assert np.array_equal(N(res["rays_d"]), O.get_rays(N(poses[:1]), intr, H, W, offset=off)[1])
    res = get_rays(poses, intr, H, W, 128, aabb=[-1, -1, -1, 1, 1, 1], min_near=0.2)
    n1, f1 = O.near_far_from_aabb(N(res["rays_o"]).reshape(-1, 3), N(res["rays_d"]).reshape(-1, 3), [-1, -1, -1, 1, 1, 1], 0.2)
    assert np.array_equal(N(res["nears"]).reshape(-1), n1) and np.array_equal(N(res["fars"]).reshape(-1), f1)
    with pytest.raises(RuntimeError):
        get_rays(poses.cpu(), intr, H, W, -1)                                           # no CPU fallback
