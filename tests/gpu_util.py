"""helpers for the -m gpu tests: numpy <-> cuda tensors, scene builders"""
import numpy as np
import torch

DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def half_from_bits(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.float16)).to(DEV)


def bits_from_half(t):
    return t.detach().cpu().numpy().view(np.uint16)


def N(t):
    return t.detach().float().cpu().numpy() if t.dtype == torch.float16 else t.detach().cpu().numpy()


def scene(C=1, bound=1.0, n_rays=512, seed=0, radius_cam=None, min_near=0.2):
    from laenerf_amd import synthetic as S
    from oracle import oracle as O
    grid = S.sphere_density_grid(cascade=C, bound=bound)
    bits = S.pack_bits_np(grid, 10.0)
    o, d = S.lego_like_rays(n_rays, seed=seed, radius=radius_cam or (3.2 if bound == 1 else 2.6))
    nears, fars = O.near_far_from_aabb(o, d, [-bound] * 3 + [bound] * 3, min_near)
    return dict(grid=grid, bits=bits, o=o, d=d, nears=nears, fars=fars, C=C, bound=bound)
