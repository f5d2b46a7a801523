"""-m gpu: randomized differential tests for the rows SURVEY 8f marks "next" (callers either side of the hot path), beside their
hand-picked cases: get_rays (nerf/utils.py:61-153), the density-grid kernels of update_extra_state / mark_untrained_grid
(nerf/renderer.py:482-649), FusedAdam against torch.optim.Adam + torch.amp.GradScaler (main_nerf.py:223, nerf/utils.py:1474-1482).
Seeded hypothesis draws (derandomised); bars as in the fixed tests: bit-exact against the oracle for rays / positions / the max-rule
grid update, the optimizer within a few ulp of one update."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import os
FUZZ_SCALE = max(1, int(os.environ.get("LAE_FUZZ_SCALE", "1")))   # LAE_FUZZ_SCALE=20: a deep one-off run (profiles/r6_deep_fuzz.txt); the default stays quick

from gpu_util import DEV, N, T

pytestmark = pytest.mark.gpu
FUZZ = dict(deadline=None, derandomize=True, database=None,
            suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large, HealthCheck.filter_too_much])


def _oracle():
    from oracle import oracle
    oracle.build()
    return oracle


def _poses(rng, B, radius):
    """camera-to-world matrices looking roughly at the origin from a sphere, with a random roll"""
    poses = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    for b in range(B):
        p = rng.standard_normal(3); p *= radius / np.linalg.norm(p)
        fwd = -p / np.linalg.norm(p)
        up = rng.standard_normal(3); up -= fwd * (up @ fwd); up /= np.linalg.norm(up)
        right = np.cross(up, fwd)
        poses[b, :3, 0], poses[b, :3, 1], poses[b, :3, 2], poses[b, :3, 3] = right, up, fwd, p
    return poses


@settings(max_examples=120 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"B": st.integers(1, 4), "H": st.integers(2, 90), "W": st.integers(2, 120), "n": st.integers(0, 3000),
                              "per_pose": st.booleans(), "offset": st.booleans(), "aabb": st.booleans(), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_get_rays(c):
    """pixel index -> ray (all pixels / one index list for every pose / one list per pose, optional sub-pixel offset), with and without
    the fused ray / box interval: bit for bit against the oracle"""
    O = _oracle()
    from laenerf_amd import _lib
    from laenerf_amd._lib import check, ptr, stream
    rng = np.random.default_rng(c["seed"])
    B, H, W = c["B"], c["H"], c["W"]
    poses = _poses(rng, B, float(rng.uniform(1.5, 4.0)))
    intr = (float(rng.uniform(0.5, 2.0) * W), float(rng.uniform(0.5, 2.0) * W), W / 2 + float(rng.uniform(-3, 3)), H / 2 + float(rng.uniform(-3, 3)))
    n = c["n"]
    inds = None
    if n > 0:
        inds = rng.integers(0, H * W, (B, n) if (c["per_pose"] and B > 1) else (n,)).astype(np.int64)
    nn = H * W if inds is None else n
    off = (float(rng.random()), float(rng.random())) if c["offset"] else None
    ro0, rd0 = O.get_rays(poses, intr, H, W, inds=inds, offset=off)
    ro = torch.empty(B, nn, 3, device=DEV); rd = torch.empty(B, nn, 3, device=DEV)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32) * np.float32(rng.uniform(0.3, 2.0))
    ab = T(aabb) if c["aabb"] else None
    nears = torch.empty(B, nn, device=DEV) if c["aabb"] else None
    fars = torch.empty(B, nn, device=DEV) if c["aabb"] else None
    ti = T(inds) if inds is not None else None
    stride = nn if (inds is not None and inds.ndim == 2) else 0
    check(_lib.load().lae_get_rays(ptr(T(poses)), B, intr[0], intr[1], intr[2], intr[3], H, W, ptr(ti), stride, nn, 0 if off is None else 1,
                                   0.0 if off is None else off[0], 0.0 if off is None else off[1], ptr(ro), ptr(rd), ptr(ab), 0.05, ptr(nears), ptr(fars),
                                   stream()), "get_rays")
    assert np.array_equal(N(ro), ro0) and np.array_equal(N(rd), rd0)
    if c["aabb"]:
        n1, f1 = O.near_far_from_aabb(ro0.reshape(-1, 3), rd0.reshape(-1, 3), aabb, 0.05)
        assert np.array_equal(N(nears).reshape(-1), n1) and np.array_equal(N(fars).reshape(-1), f1)


@settings(max_examples=80 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"H": st.sampled_from([8, 16, 32, 64]), "bound_c": st.sampled_from([0.5, 1.0, 2.0, 4.0]), "n": st.integers(1, 20000),
                              "noise": st.booleans(), "scale": st.floats(0.1, 30.0), "decay": st.sampled_from([0.5, 0.9, 0.95]), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_density_grid_positions_and_update(c):
    """cell -> jittered position + Morton index (renderer.py:580-592, 602-621) and the EMA-max update of one cascade with duplicates,
    untrained (-1) cells and negative densities (renderer.py:625-636; duplicates resolved by the maximum candidate): bit for bit"""
    O = _oracle()
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(c["seed"])
    H, n = c["H"], c["n"]
    coords = rng.integers(0, H, (n, 3)).astype(np.int32)
    noise = rng.random((n, 3), dtype=np.float32) if c["noise"] else None
    xyz, idx = rm.density_grid_positions(n, H, c["bound_c"], noise=None if noise is None else T(noise), coords=T(coords))
    rx, ri = O.density_grid_positions(n, H, c["bound_c"], noise=noise, coords=coords)
    assert np.array_equal(N(xyz), rx) and np.array_equal(N(idx), ri)
    cells = H ** 3
    grid = (rng.random(cells, dtype=np.float32) * 2).astype(np.float32)
    grid[rng.random(cells) < 0.2] = -1.0
    sig = (rng.random(n, dtype=np.float32) * 3).astype(np.float32)
    sig[rng.random(n) < 0.02] = -0.5
    tmp = torch.zeros(cells, dtype=torch.int32, device=DEV)
    g = T(grid.copy())
    rm.density_grid_update(g, T(sig), T(ri), tmp, density_scale=c["scale"], decay=c["decay"])
    assert np.array_equal(N(g), O.density_grid_update(grid, sig, ri, c["scale"], c["decay"], rule=1))
    assert int(tmp.abs().sum().item()) == 0


@settings(max_examples=40 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"B": st.integers(1, 80), "C": st.sampled_from([1, 2, 3]), "H": st.sampled_from([16, 32]), "min_near": st.sampled_from([0.05, 0.2]),
                              "close": st.booleans(), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_mark_untrained_grid(c):
    """cells no camera sees get -1 (renderer.py:482-554): equal to the oracle except within 1e-5 of a decision boundary (the oracle
    reports every cell's margin), across cascades, pose counts beyond one LDS chunk and the close-point filter"""
    O = _oracle()
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(c["seed"])
    C, H = c["C"], c["H"]
    bound = float(2 ** (C - 1))
    poses = _poses(rng, c["B"], float(rng.uniform(0.6, 2.5)) * bound)
    intr = (float(rng.uniform(40, 200)), float(rng.uniform(40, 200)), float(rng.uniform(20, 60)), float(rng.uniform(20, 60)))
    got = rm.mark_untrained_grid(torch.zeros(C, H ** 3, device=DEV), poses, intr, bound, c["min_near"], c["close"], H)
    ref, margin = O.mark_untrained_grid(np.zeros((C, H ** 3), np.float32), poses, intr, bound, c["min_near"], c["close"], H)
    diff = N(got) != ref
    assert not diff[margin > 1e-5].any(), int(diff[margin > 1e-5].sum())
    assert diff.sum() <= max(8, diff.size // 2000)


def _small_net(seed):
    from laenerf_amd.network import NeRFNetwork
    torch.manual_seed(seed)
    net = NeRFNetwork(bound=1, log2_hashmap_size=10, num_levels=8).to(DEV)
    net.encoder.embeddings.data.uniform_(-0.1, 0.1)
    return net


@settings(max_examples=60 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"lr": st.sampled_from([1e-3, 1e-2, 5e-2]), "b1": st.sampled_from([0.8, 0.9]), "b2": st.sampled_from([0.99, 0.999]),
                              "eps": st.sampled_from([1e-15, 1e-8]), "init_scale": st.sampled_from([2.0 ** 7, 2.0 ** 12, 2.0 ** 16]), "growth": st.integers(1, 4),
                              "inf_at": st.integers(-1, 7), "nan": st.booleans(), "gscale": st.sampled_from([1e-5, 1e-3, 1e-1]), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_fused_adam_against_torch_adam_and_gradscaler(c):
    """eight steps of identical gradients through FusedAdam and through torch.optim.Adam + torch.amp.GradScaler: random hyper-parameters,
    scale growth every 1-4 finite steps, one inf / NaN step (skip + backoff), gradients arriving in the fp16 accumulators or through
    plain autograd `.grad`: parameters within a few ulp of one update, moments and the loss scale equal, skipped steps counted"""
    from laenerf_amd.optim import FusedAdam
    a, b = _small_net(c["seed"] % 97), _small_net(c["seed"] % 97)
    fa = FusedAdam(a, param_groups=a.get_params(c["lr"]), betas=(c["b1"], c["b2"]), eps=c["eps"], init_scale=c["init_scale"], growth_interval=c["growth"])
    tb = torch.optim.Adam(b.get_params(c["lr"]), betas=(c["b1"], c["b2"]), eps=c["eps"])
    sc = torch.amp.GradScaler("cuda", init_scale=c["init_scale"], growth_interval=c["growth"])
    gen = torch.Generator(device=DEV).manual_seed(c["seed"])
    pa, pb = [p for g in a.get_params(0) for p in g["params"]], [p for g in b.get_params(0) for p in g["params"]]
    owners = {id(a.encoder.embeddings): a.encoder, id(a.sigma_net.weights): a.sigma_net, id(a.color_net.weights): a.color_net}
    skipped = 0
    for it in range(8):
        sc.scale(torch.zeros((), device=DEV))
        scale = sc.get_scale()
        assert fa.get_scale() == scale
        for k, (x, y) in enumerate(zip(pa, pb)):
            g = torch.randn(x.shape, device=DEV, generator=gen) * c["gscale"]
            g[torch.rand(x.shape, device=DEV, generator=gen) < 0.3] = 0
            gh = (g * scale).half()
            if it == c["inf_at"] and k == (c["seed"] % len(pa)):
                gh.view(-1)[(c["seed"] // 7) % gh.numel()] = float("nan") if c["nan"] else float("inf")
            owner = owners[id(x)]
            if (it + k) % 2 == 0:
                owner.shadow.grad_half.copy_(gh.view_as(owner.shadow.grad_half))
                owner.shadow.unreported = True
                if hasattr(owner.shadow, "mark_all_touched"):
                    owner.shadow.mark_all_touched()
            else:
                x.grad = gh.float()
            y.grad = gh.float()
        finite = all(bool(torch.isfinite(y.grad).all()) for y in pb)
        skipped += 0 if finite else 1
        fa.step()
        sc.step(tb); sc.update()
        for x, y in zip(pa, pb):
            assert torch.allclose(x, y, rtol=3e-6, atol=c["lr"] * 4e-6), (it, float((x - y).abs().max()))
        assert torch.equal(a.encoder.shadow.half, a.encoder.embeddings.detach().half())
    assert fa.steps_skipped == skipped and fa.steps_taken == 8 - skipped
    assert fa.get_scale() == sc.get_scale()
    for (p, m, v, _, _), y in zip(fa.items, pb):
        stt = tb.state[y]
        if "exp_avg" in stt:
            # (absolute floors scale with the gradients: an entry of m near a cancellation carries the rounding of its larger terms)
            assert torch.allclose(m, stt["exp_avg"], rtol=1e-5, atol=1e-6 * c["gscale"]), float((m - stt["exp_avg"]).abs().max())
            assert torch.allclose(v, stt["exp_avg_sq"], rtol=1e-5, atol=1e-6 * c["gscale"] ** 2), float((v - stt["exp_avg_sq"]).abs().max())
