"""Seeded synthetic workloads: blender-like pinhole rays and an analytic occupancy grid.
numpy only (shared by the GPU path, the tests and the CPU baseline); no dataset is needed."""
import numpy as np


def lego_like_rays(n_rays, H=800, W=800, focal=1111.1, radius=4.03 * 0.8, seed=0, n_views=16):
    """n_rays random pixels of random views on a sphere of `radius` looking at the origin
    (the reference samples `randint(0, H*W)` pixels of one view per step, nerf/utils.py:108)."""
    rng = np.random.default_rng(seed)
    view = rng.integers(0, n_views, n_rays)
    theta = rng.uniform(0.2, np.pi / 2 - 0.1, n_views)      # elevation from the pole
    phi = rng.uniform(0, 2 * np.pi, n_views)
    cam = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], -1) * radius
    fwd = -cam / np.linalg.norm(cam, axis=-1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right, axis=-1, keepdims=True)
    upv = np.cross(right, fwd)
    pix = rng.integers(0, H * W, n_rays)
    i = (pix % W).astype(np.float64) + 0.5
    j = (pix // W).astype(np.float64) + 0.5
    x = (i - W / 2) / focal
    y = -(j - H / 2) / focal
    d = fwd[view] + x[:, None] * right[view] + y[:, None] * upv[view]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return cam[view].astype(np.float32), d.astype(np.float32)


def flower_like_rays(n_rays, H=756, W=1008, focal=890.0, seed=0, n_views=34, spread=0.06, z_cam=1.5):
    """llff/flower-shaped batch (scripts/configs_llff/flower.sh: bound 2, pose scale 0.02, offset (0, 0, 1.5), min_near
    0.2): forward-facing cameras INSIDE the bound-2 box, a few centimetres apart around (0, 0, z_cam), all looking down -z
    at the scene near the origin; n_rays random pixels of random views (nerf/utils.py:108)."""
    rng = np.random.default_rng(seed)
    view = rng.integers(0, n_views, n_rays)
    cam = np.stack([rng.uniform(-spread, spread, n_views), rng.uniform(-spread, spread, n_views),
                    z_cam + rng.uniform(-0.01, 0.01, n_views)], -1)
    fwd = -cam / np.linalg.norm(cam, axis=-1, keepdims=True)            # converge on the origin (llff captures do, roughly)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right, axis=-1, keepdims=True)
    upv = np.cross(right, fwd)
    pix = rng.integers(0, H * W, n_rays)
    x = ((pix % W).astype(np.float64) + 0.5 - W / 2) / focal
    y = -((pix // W).astype(np.float64) + 0.5 - H / 2) / focal
    d = fwd[view] + x[:, None] * right[view] + y[:, None] * upv[view]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return cam[view].astype(np.float32), d.astype(np.float32)


def frame_rays(H, W, focal=None, radius=4.03 * 0.8, theta=1.0, phi=0.7):
    """all H*W rays of one view (full-frame inference, cfg4-style)"""
    focal = focal if focal is not None else 1111.1 * W / 800
    cam = np.array([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)]) * radius
    fwd = -cam / np.linalg.norm(cam)
    right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    jj, ii = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    x = (ii.reshape(-1) - W / 2) / focal
    y = -(jj.reshape(-1) - H / 2) / focal
    d = fwd[None] + x[:, None] * right[None] + y[:, None] * upv[None]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(cam, d.shape)
    return np.ascontiguousarray(o, dtype=np.float32), d.astype(np.float32)


def _morton_inverse_table(H=128):
    idx = np.arange(H ** 3, dtype=np.uint32)

    def compact(x):
        x = x & 0x49249249
        x = (x | (x >> 2)) & 0xc30c30c3
        x = (x | (x >> 4)) & 0x0f00f00f
        x = (x | (x >> 8)) & 0xff0000ff
        x = (x | (x >> 16)) & 0x0000ffff
        return x
    return compact(idx), compact(idx >> 1), compact(idx >> 2)


def sphere_density_grid(cascade=1, bound=1.0, H=128, radius=0.6, value=20.0, boxes=True):
    """density grid [C, H^3] in Morton order: `value` inside |x| < radius (and two boxes), 0 elsewhere.
    Cell centres follow the reference's update_extra_state convention (renderer.py:586-599)."""
    cx, cy, cz = _morton_inverse_table(H)
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    for c in range(cascade):
        b = min(2.0 ** c, bound)
        half = b / H
        xs = (2 * (cx.astype(np.float32) + 0.5) / H - 1) * b
        ys = (2 * (cy.astype(np.float32) + 0.5) / H - 1) * b
        zs = (2 * (cz.astype(np.float32) + 0.5) / H - 1) * b
        occ = xs * xs + ys * ys + zs * zs < (radius + half) ** 2
        if boxes:
            occ |= (np.abs(xs - 0.55) < 0.2) & (np.abs(ys + 0.3) < 0.25) & (np.abs(zs) < 0.15)
            occ |= (np.abs(xs + 0.5) < 0.12) & (np.abs(ys - 0.45) < 0.3) & (np.abs(zs - 0.4) < 0.3)
        grid[c, occ] = value
    return grid


def lego_sparse_density_grid(H=128, value=20.0, radius=0.36):
    """SURVEY 8d's second occupancy preset ("lego-like sparse", ~3 % occupied): a compact body at the origin, a base plate and two
    arms -- 3.05 % of the 128^3 cells of the bound-1 box, every one of them in view of the lego-like cameras.  On those cameras the
    frustum covers the box almost uniformly, so samples per ray follow the occupied FRACTION whatever the shape (~4.3 per percent
    at dt = 2 sqrt(3) / 1024): ~13 per ray here against ~63 on the 13 % sphere preset (the survey's "~40 per ray" does not go with its
    own 3 %)."""
    cx, cy, cz = _morton_inverse_table(H)
    xs = 2 * (cx.astype(np.float32) + 0.5) / H - 1
    ys = 2 * (cy.astype(np.float32) + 0.5) / H - 1
    zs = 2 * (cz.astype(np.float32) + 0.5) / H - 1
    occ = xs * xs + ys * ys + zs * zs < radius * radius
    occ |= (np.abs(xs) < 0.55) & (np.abs(ys + 0.28) < 0.03) & (np.abs(zs) < 0.35)
    occ |= (np.abs(xs - 0.35) < 0.05) & (np.abs(ys) < 0.3) & (np.abs(zs + 0.2) < 0.05)
    occ |= (np.abs(xs + 0.3) < 0.04) & (np.abs(ys - 0.15) < 0.35) & (np.abs(zs - 0.25) < 0.04)
    grid = np.zeros((1, H ** 3), dtype=np.float32)
    grid[0, occ] = value
    return grid


def flower_density_grid(H=128, value=20.0):
    """occupancy for the flower-shaped batches (bound 2 -> 2 cascades, renderer.py:74): a blob at the origin (cascade 0
    and 1), two petals, and a wall behind it at z in [-1.7, -1.4] that only the outer cascade can hold."""
    cx, cy, cz = _morton_inverse_table(H)
    grid = np.zeros((2, H ** 3), dtype=np.float32)
    for c in range(2):
        b = float(2 ** c)
        half = b / H
        xs = (2 * (cx.astype(np.float32) + 0.5) / H - 1) * b
        ys = (2 * (cy.astype(np.float32) + 0.5) / H - 1) * b
        zs = (2 * (cz.astype(np.float32) + 0.5) / H - 1) * b
        occ = xs * xs + ys * ys + zs * zs < (0.45 + half) ** 2
        occ |= (np.abs(xs - 0.5) < 0.25) & (np.abs(ys - 0.2) < 0.08) & (np.abs(zs - 0.1) < 0.3)
        occ |= (np.abs(xs + 0.4) < 0.08) & (np.abs(ys + 0.5) < 0.3) & (np.abs(zs) < 0.25)
        occ |= (zs > -1.7) & (zs < -1.4) & (np.abs(xs) < 1.8) & (np.abs(ys) < 1.8)
        grid[c, occ] = value
    return grid


def pack_bits_np(grid, thresh):
    bits = (grid.reshape(-1, 8) > thresh).astype(np.uint8)
    return (bits << np.arange(8, dtype=np.uint8)).sum(axis=1).astype(np.uint8)
