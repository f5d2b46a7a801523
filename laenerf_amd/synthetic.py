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


def pack_bits_np(grid, thresh):
    bits = (grid.reshape(-1, 8) > thresh).astype(np.uint8)
    return (bits << np.arange(8, dtype=np.uint8)).sum(axis=1).astype(np.uint8)
