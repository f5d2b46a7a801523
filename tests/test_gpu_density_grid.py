"""-m gpu: occupancy-grid maintenance kernels, the `run` path (R3) and the nn.Linear network (A15)
against the oracle and the vectors recorded from the reference's own Python (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import golden
from gpu_util import DEV, N, T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,bound_c", [(16, 1.0), (128, 2.0)])
def test_positions_bit_exact(O, H, bound_c):
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(H)
    n = H ** 3
    noise = rng.random((n, 3), dtype=np.float32)
    xyz, idx = rm.density_grid_positions(n, H, bound_c, noise=T(noise))
    rx, ri = O.density_grid_positions(n, H, bound_c, noise=noise)
    assert np.array_equal(N(xyz), rx) and np.array_equal(N(idx), ri)
    coords = rng.integers(0, H, (3000, 3)).astype(np.int32)
    xyz, idx = rm.density_grid_positions(3000, H, bound_c, noise=T(noise[:3000]), coords=T(coords))
    rx, ri = O.density_grid_positions(3000, H, bound_c, noise=noise[:3000], coords=coords)
    assert np.array_equal(N(xyz), rx) and np.array_equal(N(idx), ri)
    xyz, idx = rm.density_grid_positions(n, H, bound_c)                  # no jitter: cell centres
    rx, _ = O.density_grid_positions(n, H, bound_c)
    assert np.array_equal(N(xyz), rx)


@pytest.mark.parametrize("H,bound_c,frac", [(32, 1.0, 0.1), (128, 2.0, 0.03), (16, 1.0, 0.0), (16, 1.0, 1.0)])
def test_partial_sweep_selection_on_the_device_equals_the_reference_statements(O, H, bound_c, frac):
    """update_extra_state's partial sweep (nerf/renderer.py:600-621): `occ = nonzero(grid > 0); occ = occ[randint(len(occ), n)];
    coords = cat([rand_coords, morton3D_invert(occ)])` -- the reference's statements on torch tensors (with one host read) against
    the device-side selection (lae_density_grid_partial_positions: count / scan / compact / draw, no host read), fed with the SAME
    draws (u = (rank + 0.5) / K picks occupied cell number `rank`): positions and Morton indices bit for bit, with no cell occupied
    (the reference then keeps the n random points: the device marks the other half -1, which the update skips) and with all."""
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(H + int(100 * frac))
    cells, n = H ** 3, max(H ** 3 // 4, 64)
    grid = np.where(rng.random(cells) < frac, rng.random(cells, dtype=np.float32) + 0.01, -rng.random(cells, dtype=np.float32)).astype(np.float32)
    if frac > 0:
        grid[rng.integers(0, cells)] = 0.0                                 # zero is not occupied (strict >)
    g = T(grid)
    coords_r = rng.integers(0, H, (n, 3)).astype(np.int32)
    noise = rng.random((2 * n, 3), dtype=np.float32)
    occ = torch.nonzero(g > 0).squeeze(-1)
    K = int(occ.numel())
    ranks = rng.integers(0, max(K, 1), n)
    u = ((ranks + 0.5) / max(K, 1)).astype(np.float32)
    xyz, idx = rm.density_grid_partial_positions(g, T(coords_r), T(u), H, bound_c, noise=T(noise))
    assert xyz.shape == (2 * n, 3) and idx.shape == (2 * n,)
    if K == 0:
        rx, ri = rm.density_grid_positions(n, H, bound_c, noise=T(noise[:n]), coords=T(coords_r))
        assert torch.equal(xyz[:n], rx) and torch.equal(idx[:n], ri) and bool((idx[n:] == -1).all())
    else:
        coords = torch.cat([T(coords_r), rm.morton3D_invert(occ[T(ranks.astype(np.int64))].int())], dim=0)
        rx, ri = rm.density_grid_positions(2 * n, H, bound_c, noise=T(noise), coords=coords)
        assert torch.equal(idx, ri) and torch.equal(xyz, rx)
        assert bool((g[idx[n:].long()] > 0).all())                         # every drawn cell is occupied
    # u at the ends of [0, 1): first and last occupied cell, never out of range
    if K > 0:
        ue = np.zeros(n, np.float32); ue[1::2] = np.nextafter(np.float32(1.0), np.float32(0.0))
        _, idx2 = rm.density_grid_partial_positions(g, T(coords_r), T(ue), H, bound_c)
        assert int(idx2[n]) == int(occ[0]) and int(idx2[n + 1]) == int(occ[-1])


@pytest.mark.parametrize("H", [128, 8])
def test_partial_sweep_sorted_draws_made_on_the_device(H):
    """rnd path of lae_density_grid_partial_positions: both halves drawn as SORTED i.i.d. uniforms from partial sums of exponentials
    (no sort, no host read).  Against a float64 numpy evaluation of the same order statistics fed through the explicit (coords, u)
    path: the same cells up to the summation order of the partial sums (a draw within rounding of a cell boundary may land next
    door: < 0.1 % of the points, never further than one code / rank); cells in Morton order, occupied draws occupied, and the
    sample is uniform (Kolmogorov distance)."""
    from laenerf_amd import raymarching as rm
    # H = 8: n_chunks + 64 + cells scratch words is an odd count -- the double sums behind them must still be 8-byte aligned
    bound_c = 2.0
    cells, n = H ** 3, H ** 3 // 4
    rng = np.random.default_rng(5)
    grid = np.where(rng.random(cells) < 0.07, 1.0, -1.0).astype(np.float32)
    g = T(grid)
    occ = np.nonzero(grid > 0)[0]
    K = occ.size
    rnd = rng.random((2, n + 1)).astype(np.float32)
    noise = rng.random((2 * n, 3), dtype=np.float32)
    xyz, idx = rm.density_grid_partial_positions(g, None, None, H, bound_c, noise=T(noise), rnd=T(rnd))
    idx = N(idx)
    c = np.cumsum(-np.log1p(-rnd.astype(np.float64)), axis=1)
    u = np.minimum(c[:, :n] / c[:, n:], 1.0 - 2.0 ** -53)                             # the draws stay in double (fp32 cannot address 2^24+ cells / ranks)
    codes = np.minimum((u[0] * cells).astype(np.int64), cells - 1)
    ranks = np.minimum((u[1] * K).astype(np.int64), K - 1)
    d0, d1 = np.abs(idx[:n].astype(np.int64) - codes), np.abs(np.searchsorted(occ, idx[n:]) - ranks)
    assert (d0 > 0).mean() < (1e-3 if H == 128 else 2e-2) and d0.max() <= 1 and (d1 > 0).mean() < (1e-3 if H == 128 else 2e-2) and d1.max() <= 1
    assert (np.diff(idx[:n]) >= 0).all() and (np.diff(idx[n:]) >= 0).all()            # Morton order within each half
    assert (grid[idx[n:]] > 0).all()
    ks = np.abs((np.arange(n) + 0.5) / n - (idx[:n] + 0.5) / cells).max()               # empirical vs uniform CDF
    assert ks < 4.0 / np.sqrt(n), ks
    # positions of the points whose cells agree: the explicit path's arithmetic, bit for bit
    assert (d0 == 0).mean() > 0.9
    same = np.nonzero((d0 == 0))[0][:50000]
    coords = rm.morton3D_invert(T(codes[same].astype(np.int32)))
    rx, ri = rm.density_grid_positions(same.size, H, bound_c, noise=T(noise[same]), coords=coords)
    assert np.array_equal(N(xyz)[same], N(rx)) and np.array_equal(idx[same], N(ri))


def test_update_is_exact_max_rule_and_leaves_scratch_clean(O):
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(3)
    cells, n = 32 ** 3, 40000
    grid = rng.random(cells, dtype=np.float32) * 2
    grid[rng.random(cells) < 0.2] = -1.0                                   # untrained cells stay untouched
    sig = (rng.random(n, dtype=np.float32) * 3).astype(np.float32)
    sig[::97] = -0.5                                                       # negative densities are ignored (tmp >= 0 test)
    idx = rng.integers(0, cells, n).astype(np.int32)                       # plenty of duplicates
    tmp = torch.zeros(cells, dtype=torch.int32, device=DEV)
    g = T(grid.copy())
    rm.density_grid_update(g, T(sig), T(idx), tmp, density_scale=1.5, decay=0.9)
    assert np.array_equal(N(g), O.density_grid_update(grid, sig, idx, 1.5, 0.9, rule=1))
    assert int(tmp.abs().sum().item()) == 0
    g2 = T(grid.copy())
    rm.density_grid_update(g2, T(sig), T(idx), tmp, density_scale=1.5, decay=0.9)   # deterministic
    assert torch.equal(g, g2)


def test_mark_untrained_grid_vs_oracle_and_reference():
    from oracle import oracle as O
    from laenerf_amd import raymarching as rm
    g = golden("density_grid")
    H = int(g["H"])
    got = rm.mark_untrained_grid(torch.zeros(2, H ** 3, device=DEV), g["poses"], g["intrinsics"], float(g["bound"]), 0.2, False, H)
    ref, margin = O.mark_untrained_grid(np.zeros((2, H ** 3), np.float32), g["poses"], g["intrinsics"], float(g["bound"]), 0.2, False, H)
    assert np.array_equal(N(got), ref)
    assert not (N(got) != g["grid_marked"])[margin > 1e-5].any()
    # 128^3, more cameras than one LDS chunk, close-point filter on
    rng = np.random.default_rng(1)
    poses = np.tile(np.eye(4, dtype=np.float32), (300, 1, 1))
    poses[:, :3, 3] = rng.standard_normal((300, 3)) * 0.8
    q, _ = np.linalg.qr(rng.standard_normal((300, 3, 3)))
    poses[:, :3, :3] = q
    got = rm.mark_untrained_grid(torch.zeros(1, 128 ** 3, device=DEV), poses, (300.0, 310.0, 64.0, 60.0), 1.0, 0.25, True, 128)
    ref, margin = O.mark_untrained_grid(np.zeros((1, 128 ** 3), np.float32), poses, (300.0, 310.0, 64.0, 60.0), 1.0, 0.25, True, 128)
    diff = N(got) != ref
    assert not diff[margin > 1e-5].any() and diff.sum() < 50 and 0 < (ref < 0).sum() < 128 ** 3


class ReplayRNG:
    def __init__(self, rands, ints):
        self.rands, self.ints = list(rands), list(ints)

    def rand(self, n, k):
        r = self.rands.pop(0); assert r.shape == (n, k); return torch.from_numpy(r)

    def randint(self, high, shape):
        r = self.ints.pop(0); assert tuple(r.shape) == tuple(shape) and r.max() < high; return torch.from_numpy(r)


def test_update_extra_state_matches_reference_python():
    """the whole refresh (full sweep, then partial sweep) under the reference's recorded random draws"""
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    g = golden("density_grid")
    H, bound = int(g["H"]), int(g["bound"])
    net = NeRFNetwork(bound=bound, log2_hashmap_size=10).to(DEV)
    assert net.encoder.embeddings.shape == g["table"].shape
    net.encoder.embeddings.data = T(g["table"]); net.sigma_net.weights.data = T(g["sigma_w"]); net.color_net.weights.data = T(g["color_w"])
    r = NeRFRenderer(net, bound=bound, grid_size=H, density_thresh=10).to(DEV)
    assert r.mark_untrained_grid(g["poses"], g["intrinsics"]) == int((g["grid_marked"] < 0).sum())
    net.train()
    with torch.autocast("cuda", dtype=torch.float16):
        r.update_extra_state(rng=ReplayRNG([g["full_noise0"], g["full_noise1"]], []))
    assert r.iter_density == 1 and net.training
    got = N(r.density_grid)
    assert np.array_equal(got < 0, g["full_grid"] < 0)
    assert np.allclose(got, g["full_grid"], rtol=2e-2, atol=1e-3)        # fp16 table interpolation + MFMA summation order
    assert abs(r.mean_density - float(g["full_mean"])) < 5e-3
    bits = np.unpackbits(N(r.density_bitfield)); ref_bits = np.unpackbits(g["full_bitfield"])
    assert (bits != ref_bits).mean() < 0.01
    # partial sweep: start from the reference's grid so both sides draw the same occupied cells
    r.density_grid.copy_(T(g["full_grid"]))
    r.iter_density = 16
    r.local_step = 3
    r.step_counter[:3, 0] = torch.tensor([1000, 1200, 1100], dtype=torch.int32, device=DEV)
    with torch.autocast("cuda", dtype=torch.float16):
        r.update_extra_state(rng=ReplayRNG([g["part_noise0"], g["part_noise1"]],
                                           [g["part_coords0"], g["part_pick0"], g["part_coords1"], g["part_pick1"]]))
    got = N(r.density_grid)
    # cells hit once agree with the reference; cells hit repeatedly hold the maximum candidate (documented rule)
    from oracle import oracle as O
    for cas in range(2):
        occ = np.nonzero(g["full_grid"][cas] > 0)[0][g[f"part_pick{cas}"]]
        coords = np.concatenate([g[f"part_coords{cas}"].astype(np.int32), O.morton3D_invert(occ.astype(np.int32))])
        idx = O.morton3D(coords)
        hits = np.bincount(idx, minlength=H ** 3)
        assert np.allclose(got[cas][hits <= 1], g["part_grid"][cas][hits <= 1], rtol=2e-2, atol=1e-3)
        exp = O.density_grid_update(g["full_grid"][cas], g[f"part_sigma{cas}"], idx, 1.0, 0.95, rule=1)
        assert np.allclose(got[cas], exp, rtol=2e-2, atol=1e-3)
    assert r.mean_count == int(g["part_mean_count"]) and r.local_step == 0


class AnalyticModel(torch.nn.Module):
    """the analytic field of tests/golden/make_golden.py (AnalyticField)"""

    def density(self, x):
        return {"sigma": 40.0 * torch.exp(-6.0 * (x * x).sum(-1))}

    def color(self, x, d, mask=None, **kw):
        rgb = torch.sigmoid(torch.stack([3 * x[:, 0] + d[:, 1], 2 * x[:, 1] - d[:, 2], x[:, 2] * 4 + d[:, 0]], -1))
        return rgb if mask is None else rgb * mask[:, None]


@pytest.mark.parametrize("name", ["run_path", "run_upsample"])
def test_run_path_matches_reference(name):
    from laenerf_amd.renderer import NeRFRenderer
    g = golden(name)
    r = NeRFRenderer(AnalyticModel(), bound=1).to(DEV).eval()
    ups = int(g["upsample_steps"]) if "upsample_steps" in g else 0
    with torch.no_grad():
        res = r.run(T(g["rays_o"])[None], T(g["rays_d"])[None], num_steps=int(g["num_steps"]), upsample_steps=ups, bg_color=1)
    tol = 2e-5 if ups == 0 else 5e-4        # resampling amplifies last-ulp differences of exp/cumsum between CPU and GPU
    assert np.abs(N(res["image"][0]) - g["image"]).max() < tol
    dep = N(res["depth"][0])
    assert np.array_equal(np.isnan(dep), np.isnan(g["depth"]))            # rays missing the box: (z - near) / (far - near) = 0/0
    assert np.nanmax(np.abs(dep - g["depth"])) < tol
    assert np.abs(N(res["weights_sum"]) - g["weights_sum"]).max() < tol


def test_linear_network_matches_reference_chain():
    """A15: NeRFNetworkLinear == nerf/network.py on the recorded weights (3-layer sigma / 4-layer colour variant)"""
    from laenerf_amd.network import NeRFNetworkLinear
    g = golden("mlp_chain")
    net = NeRFNetworkLinear(num_layers=3, num_layers_color=4, log2_hashmap_size=10).to(DEV)
    net.encoder.embeddings.data = T(g["table"])
    for layer, w in zip(net.sigma_net, [g["sigma_w"], g["sigma_w1"], g["sigma_w2"]]):
        layer.weight.data = T(w)
    for layer, w in zip(net.color_net, [g["color_w0"], g["color_w1"], g["color_w2"], g["color_w3"]]):
        layer.weight.data = T(w)
    with torch.no_grad():
        sigma, color = net(T(g["x"]), T(g["d"]))
        assert np.allclose(N(sigma), g["sigma"], rtol=1e-4) and np.abs(N(color) - g["color"]).max() < 1e-5
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, color = net(T(g["x"]), T(g["d"]))
        assert np.allclose(N(sigma.float()), g["sigma"], rtol=1e-2) and np.abs(N(color.float()) - g["color"]).max() < 3e-3
        d = net.density(T(g["x"]))
        assert d["geo_feat"].shape == (512, 15)
        mask = torch.zeros(512, dtype=torch.bool, device=DEV); mask[::3] = True
        c = net.color(T(g["x"]), T(g["d"]), mask=mask, geo_feat=d["geo_feat"])
        assert np.abs(N(c)[::3] - g["color"][::3]).max() < 1e-5 and float(c[~mask].abs().sum()) == 0


def test_density_head_equals_full_head():
    from laenerf_amd.ffmlp import nerf_density, nerf_head
    rng = np.random.default_rng(0)
    M = 1000                                                           # padded to 1008 inside
    enc = T((rng.standard_normal((M, 32)) * 0.3).astype(np.float16))
    ws = T(rng.uniform(-0.2, 0.2, 64 * 112).astype(np.float32)); wc = T(rng.uniform(-0.2, 0.2, 64 * 176).astype(np.float32))
    sig, h = nerf_density(enc, ws)
    d = torch.nn.functional.normalize(torch.randn(1008, 3, device=DEV), dim=-1)
    with torch.no_grad():
        s2, _ = nerf_head(torch.cat([enc, enc.new_zeros(8, 32)]), d, ws, wc)
    assert torch.equal(sig, s2[:M]) and h.shape == (M, 16)
    assert torch.equal(nerf_density(enc, ws, 2.0, want_geo_feat=False)[0], 2.0 * sig)


def test_renderer_loads_reference_layout_checkpoints():
    """the reference's NeRFNetwork IS its renderer, so its checkpoints name `encoder.embeddings`, `sigma_net.weights`, ... next to
    `density_grid`, `density_bitfield`, `aabb_*`, `step_counter` (nerf/utils.py:1587 saves {'model': state_dict}); both that
    layout and this repo's (`model.` prefix) load, and state_dict(reference_layout=True) writes the reference's"""
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.optim import FusedAdam
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(0)
    a = NeRFRenderer(NeRFNetwork(bound=1, log2_hashmap_size=12), bound=1).to(DEV)
    a.model.encoder.embeddings.data.uniform_(-0.3, 0.3)
    a.density_grid.uniform_(0, 20); a.density_bitfield.random_(0, 255); a.step_counter.random_(0, 1000)
    ref_sd = a.state_dict(reference_layout=True)
    assert {"encoder.embeddings", "encoder.offsets", "sigma_net.weights", "color_net.weights", "density_grid", "density_bitfield",
            "aabb_train", "aabb_infer", "step_counter"} <= set(ref_sd.keys()) and not any(k.startswith("model.") for k in ref_sd)
    for payload in (ref_sd, {"model": ref_sd, "epoch": 3}, a.state_dict()):
        torch.manual_seed(1)
        b = NeRFRenderer(NeRFNetwork(bound=1, log2_hashmap_size=12), bound=1).to(DEV)
        opt = FusedAdam(b.model, param_groups=b.model.get_params(1e-2))          # fp16 shadows exist before the load
        b.load_state_dict(payload)
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka
        assert torch.equal(b.model.encoder.shadow.half, a.model.encoder.embeddings.detach().half())    # shadow followed the load
        del opt
