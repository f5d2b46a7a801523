"""-m gpu: randomized differential tests of the operators against the oracle (oracle/lae_oracle.c), beside the hand-picked cases
of test_gpu_raymarching.py / test_gpu_encoders.py: seeded hypothesis draws (derandomised) over the parameters the reference's
operators take -- ray counts down to 1, cascades 1-4 and bounds that are not powers of two, dt_gamma on / off, max_steps, occupancy
patterns from empty to full, exact-fit and overflowing sample buffers, encoder shapes D x C x L x table size x grid type x
align_corners x interpolation in fp32 and fp16, out-of-range inputs, ragged compositing segments.

Bars: bit-exact for integers, indices, offsets, sample positions and encoder outputs (fp32 and fp16); the compositing sums within
the tolerances of test_gpu_raymarching.py (wave-level scans associate differently from the serial loop).
Reference: raymarching/src/raymarching.cu:91-156 (near_far), :214-300 (morton, packbits), :311-490 (march_rays_train), :500-693
(composite_rays_train), :700-805 / :948-1035 (march_rays / composite_rays); gridencoder/src/gridencoder.cu:87-245."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import os
FUZZ_SCALE = max(1, int(os.environ.get("LAE_FUZZ_SCALE", "1")))   # LAE_FUZZ_SCALE=20: a deep one-off run (profiles/r6_deep_fuzz.txt); the default stays quick

from gpu_util import DEV, N, T, bits_from_half, half_from_bits

pytestmark = pytest.mark.gpu
FUZZ = dict(deadline=None, derandomize=True, database=None,
            suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large, HealthCheck.filter_too_much])


def _oracle():
    from oracle import oracle
    oracle.build()
    return oracle


def _bits(kind, C, bound, p, seed):
    from laenerf_amd import synthetic as S
    n_bytes = C * 128 ** 3 // 8
    if kind == "empty":
        return np.zeros(n_bytes, np.uint8)
    if kind == "full":
        return np.full(n_bytes, 255, np.uint8)
    if kind == "random":
        return np.packbits(np.random.default_rng(seed).random(n_bytes * 8) < p, bitorder="little")
    return S.pack_bits_np(S.sphere_density_grid(cascade=C, bound=float(bound), boxes=(kind == "boxes")), 10.0)


def _rays(n, seed, bound, inside):
    """lego-like cameras outside the box, or origins INSIDE it (the mip360 / llff case) with random directions"""
    from laenerf_amd import synthetic as S
    rng = np.random.default_rng(seed)
    if inside:
        o = rng.uniform(-0.8 * bound, 0.8 * bound, (n, 3)).astype(np.float32)
        d = rng.standard_normal((n, 3)).astype(np.float32)
        d[rng.random(n) < 0.05, 0] = 0.0                       # axis-parallel components: infinite slabs in near_far
        d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-6)
        return o, d.astype(np.float32)
    return S.lego_like_rays(n, seed=seed, radius=float(rng.uniform(1.2, 3.0)) * max(bound, 1.0))


march_draw = st.fixed_dictionaries({
    "n": st.one_of(st.integers(1, 70), st.integers(71, 900)),
    "C": st.sampled_from([1, 1, 2, 3, 4]),
    "bound_frac": st.sampled_from([1.0, 1.0, 0.75]),             # 0.75: a bound that is not a power of two (cascade = 1 + ceil(log2 bound))
    "dtg": st.sampled_from([0.0, 0.0, 1 / 256, 1 / 128]),
    "max_steps": st.sampled_from([32, 256, 1024]),
    "occ": st.sampled_from(["sphere", "boxes", "random", "random", "empty", "full"]),
    "p": st.floats(0.0, 0.6),
    "inside": st.booleans(),
    "perturb": st.booleans(),
    "fit": st.sampled_from(["full", "exact", "half"]),
    "seed": st.integers(0, 2 ** 20),
})


@settings(max_examples=300 * FUZZ_SCALE, **FUZZ)
@given(march_draw)
def test_fuzz_ray_box_and_training_march(c):
    """near_far_from_aabb + march_rays_train: nears / fars, rays (id, offset, count), counter, rows_end, every sample row and the
    zero fill of the rows no ray owns -- bit for bit"""
    O = _oracle()
    from laenerf_amd.backend import raymarching_backend as B
    C, n = c["C"], c["n"]
    bound = float(2 ** (C - 1)) * c["bound_frac"]
    if C == 1:
        bound = 1.0 * c["bound_frac"]
    Cc = max(1, 1 + int(np.ceil(np.log2(bound))))             # renderer.py:74
    bits = _bits(c["occ"], Cc, bound, c["p"], c["seed"])
    o, d = _rays(n, c["seed"], bound, c["inside"])
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    n0, f0 = O.near_far_from_aabb(o, d, aabb, 0.2)
    nears, fars = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    B.near_far_from_aabb(T(o), T(d), T(aabb), n, 0.2, nears, fars)
    assert np.array_equal(N(nears), n0) and np.array_equal(N(fars), f0)
    noises = np.random.default_rng(c["seed"] + 1).random(n).astype(np.float32) if c["perturb"] else np.zeros(n, np.float32)
    ms = c["max_steps"]
    full = O.march_rays_train(o, d, bound, bits, Cc, 128, n0, f0, noises, dt_gamma=c["dtg"], max_steps=ms)
    total = int(full[4][0])
    M = {"full": n * ms, "exact": max(total, 1), "half": max(total // 2, 1)}[c["fit"]]
    ref = O.march_rays_train(o, d, bound, bits, Cc, 128, n0, f0, noises, M=M, dt_gamma=c["dtg"], max_steps=ms)
    xyzs = torch.full((M, 3), float("nan"), device=DEV); dirs = torch.full((M, 3), float("nan"), device=DEV)
    deltas = torch.full((M, 2), float("nan"), device=DEV)
    rays = torch.empty(n, 3, dtype=torch.int32, device=DEV); counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    rows_end = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    B.march_rays_train(T(o), T(d), T(bits), bound, c["dtg"], ms, n, Cc, 128, M, T(n0), T(f0), xyzs, dirs, deltas, rays, counter, T(noises), rows_end)
    assert np.array_equal(N(counter), ref[4]), (N(counter), ref[4])
    assert np.array_equal(N(rays), ref[3])
    assert np.array_equal(N(xyzs), ref[0]) and np.array_equal(N(dirs), ref[1]) and np.array_equal(N(deltas), ref[2])
    fit = ref[3][(ref[3][:, 2] > 0) & (ref[3][:, 1] + ref[3][:, 2] <= M)]
    assert int(rows_end.item()) == (int((fit[:, 1] + fit[:, 2]).max()) if len(fit) else 0)


@settings(max_examples=100 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"n": st.integers(1, 5000), "thresh": st.floats(-1.0, 30.0), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_morton_and_packbits(c):
    O = _oracle()
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(c["seed"])
    coords = rng.integers(0, 128, (c["n"], 3)).astype(np.int32)
    idx = rm.morton3D(T(coords))
    assert np.array_equal(N(idx), O.morton3D(coords))
    assert np.array_equal(N(rm.morton3D_invert(idx)), coords)
    grid = rng.uniform(-5, 40, (1, 8 * ((c["n"] + 7) // 8))).astype(np.float32)
    grid[0, : min(8, grid.shape[1])] = np.float32(c["thresh"])   # values exactly AT the threshold (strict >)
    assert np.array_equal(N(rm.packbits(T(grid), float(np.float32(c["thresh"])))), O.packbits(grid, np.float32(c["thresh"])))


@settings(max_examples=200 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"n": st.integers(1, 400), "mean_len": st.sampled_from([0.5, 3.0, 40.0, 200.0]), "sig": st.sampled_from([0.1, 5.0, 200.0]),
                              "T_thresh": st.sampled_from([1e-4, 1e-2]), "shuffle": st.booleans(), "overflow": st.booleans(),
                              "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_composite_rays_train(c):
    """ragged segments (zero-length rays, one-sample rays, rays longer than a wavefront), shuffled ray rows, segments that run past M"""
    O = _oracle()
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(c["seed"])
    n = c["n"]
    counts = rng.poisson(c["mean_len"], n).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(counts[:-1])]).astype(np.int32)
    total = int(counts.sum())
    M = max(total, 1)
    if c["overflow"]:
        M = max(total // 2, 1)                                 # rays whose segment does not fit are skipped (raymarching.cu:521, 624)
    rays = np.stack([np.arange(n, dtype=np.int32), offs, counts], 1)
    if c["shuffle"]:
        rays = rays[rng.permutation(n)]
    sig = (rng.exponential(c["sig"], M)).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    dl = np.stack([np.full(M, 0.0034, np.float32), rng.uniform(0.003, 0.05, M).astype(np.float32)], 1)
    ws0, dep0, img0 = O.composite_rays_train_forward(sig, rgb, dl, rays, c["T_thresh"])
    s, col = T(sig).requires_grad_(), T(rgb).requires_grad_()
    ws, dep, img = rm.composite_rays_train(s, col, T(dl), T(rays), c["T_thresh"])
    assert np.allclose(N(ws), ws0, atol=2e-6) and np.allclose(N(img), img0, atol=2e-6) and np.allclose(N(dep), dep0, atol=2e-5, rtol=1e-5)
    gws, gimg = rng.standard_normal(n).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32)
    torch.autograd.backward([ws, img], [T(gws), T(gimg)])
    gs0, gc0 = O.composite_rays_train_backward(gws, gimg, sig, rgb, dl, rays, ws0, img0, c["T_thresh"])
    assert np.allclose(N(s.grad), gs0, atol=3e-5, rtol=1e-4) and np.allclose(N(col.grad), gc0, atol=1e-6, rtol=1e-5)


grid_draw = st.fixed_dictionaries({
    "D": st.sampled_from([2, 3, 3, 3]), "C": st.sampled_from([1, 2, 2, 4, 8]), "L": st.integers(1, 16),
    "base": st.sampled_from([2, 4, 16]), "pls": st.floats(1.15, 2.0), "T_log2": st.integers(8, 16),
    "gridtype": st.sampled_from([0, 0, 1]), "align": st.booleans(), "interp": st.sampled_from([0, 1]),
    "B": st.one_of(st.integers(1, 130), st.integers(131, 6000)), "half": st.booleans(), "oob": st.booleans(),
    "seed": st.integers(0, 2 ** 20),
})


@settings(max_examples=400 * FUZZ_SCALE, **FUZZ)
@given(grid_draw)
def test_fuzz_grid_encode_forward(c):
    """hash / tiled grids of every supported shape, fp32 and fp16 tables, both output layouts: bit-identical to the oracle"""
    O = _oracle()
    from laenerf_amd.backend import gridencoder_backend as G
    D, C, L, B = c["D"], c["C"], c["L"], c["B"]
    if c["half"] and C % 2:
        C = 2                                                  # odd level_dim stays float (grid.py:41-44)
    offsets, pls = O.grid_offsets(input_dim=D, num_levels=L, level_dim=C, per_level_scale=c["pls"], base_resolution=c["base"],
                                  log2_hashmap_size=c["T_log2"], align_corners=c["align"])
    rng = np.random.default_rng(c["seed"])
    table = rng.uniform(-0.5, 0.5, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.random((B, D)).astype(np.float32)
    x[rng.random(B) < 0.05] = np.float32(1.0)                   # the upper edge is inside (<= 1)
    x[rng.random(B) < 0.05] = np.float32(0.0)
    if c["oob"]:
        x[rng.random(B) < 0.1, 0] = np.float32(1.0000001)       # out of range: zero output (gridencoder.cu:118-135)
        x[rng.random(B) < 0.05, D - 1] = np.float32(-1e-7)
    if c["half"]:
        th = O.to_f16_bits(table)
        ref, _ = O.grid_encode_forward(x, th, offsets, pls, c["base"], gridtype=c["gridtype"], align_corners=c["align"], interp=c["interp"], f16=True)
        out = torch.full((L, B, C), 7.0, device=DEV, dtype=torch.half)
        G.grid_encode_forward(T(x), half_from_bits(th), T(offsets), out, B, D, C, L, np.log2(pls), c["base"], None, c["gridtype"], c["align"], c["interp"])
        assert np.array_equal(bits_from_half(out), ref)
        out2 = torch.full((B, L * C), 7.0, device=DEV, dtype=torch.half)
        G.grid_encode_forward(T(x), half_from_bits(th), T(offsets), out2, B, D, C, L, np.log2(pls), c["base"], None, c["gridtype"], c["align"], c["interp"], blc=True)
        assert np.array_equal(bits_from_half(out2).reshape(B, L, C).transpose(1, 0, 2), ref)
    else:
        ref, ref_dd = O.grid_encode_forward(x, table, offsets, pls, c["base"], calc_dy_dx=True, gridtype=c["gridtype"], align_corners=c["align"], interp=c["interp"])
        out = torch.full((L, B, C), 7.0, device=DEV); dd = torch.empty(B, L * D * C, device=DEV)
        G.grid_encode_forward(T(x), T(table), T(offsets), out, B, D, C, L, np.log2(pls), c["base"], dd, c["gridtype"], c["align"], c["interp"])
        assert np.array_equal(N(out), ref)
        assert np.allclose(N(dd), ref_dd, rtol=1e-5, atol=1e-6)


@settings(max_examples=150 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"C": st.sampled_from([1, 2, 4]), "L": st.integers(1, 12), "T_log2": st.integers(8, 15), "pls": st.floats(1.2, 2.0),
                              "B": st.integers(1, 4000), "gridtype": st.sampled_from([0, 1]), "align": st.booleans(), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_grid_encode_backward_fp32(c):
    """fp32 table gradient against the oracle's sequential sums (different summation order: relative 1e-5 of the entry's absolute sum)"""
    O = _oracle()
    from laenerf_amd.backend import gridencoder_backend as G
    C, L, B = c["C"], c["L"], c["B"]
    offsets, pls = O.grid_offsets(input_dim=3, num_levels=L, level_dim=C, per_level_scale=c["pls"], base_resolution=16,
                                  log2_hashmap_size=c["T_log2"], align_corners=c["align"])
    rng = np.random.default_rng(c["seed"])
    x = rng.random((B, 3)).astype(np.float32)
    g = (rng.standard_normal((L, B, C)) * 0.1).astype(np.float32)
    ref, _ = O.grid_encode_backward(g, x, (int(offsets[-1]), C), offsets, pls, 16, gridtype=c["gridtype"], align_corners=c["align"])
    ge = torch.zeros(int(offsets[-1]), C, device=DEV)
    emb = torch.zeros(int(offsets[-1]), C, device=DEV)
    G.grid_encode_backward(T(g), T(x), emb, T(offsets), ge, B, 3, C, L, np.log2(pls), 16, None, None, c["gridtype"], c["align"], 0)
    got = N(ge)
    scale = max(float(np.abs(ref).max()), 1e-6)
    assert np.abs(got - ref).max() <= 2e-5 * scale + 1e-7, float(np.abs(got - ref).max() / scale)
    assert np.array_equal(got != 0, ref != 0) or float(((got != 0) ^ (ref != 0)).mean()) < 1e-4


infer_draw = st.fixed_dictionaries({
    "n": st.integers(1, 600), "C": st.sampled_from([1, 2, 3]), "occ": st.sampled_from(["sphere", "random", "full", "empty"]), "p": st.floats(0.0, 0.5),
    "n_step": st.integers(1, 8), "dtg": st.sampled_from([0.0, 1 / 128]), "max_steps": st.sampled_from([64, 1024]),
    "alive_frac": st.floats(0.05, 1.0), "edit": st.booleans(), "perturb": st.booleans(), "T_thresh": st.sampled_from([1e-4, 1e-2]),
    "seed": st.integers(0, 2 ** 20),
})


@settings(max_examples=300 * FUZZ_SCALE, **FUZZ)
@given(infer_draw)
def test_fuzz_inference_operators(c):
    """one iteration of the reference's inference loop from a random state: march_rays(+distill) rows bit for bit, composite_rays(+distill)
    alive marks and rays_t bit for bit, accumulators within 3e-5 (raymarching.cu:700-805, 948-1035, 1037-1142)"""
    O = _oracle()
    from laenerf_amd import raymarching as rm
    from laenerf_amd.backend import raymarching_backend as B
    rng = np.random.default_rng(c["seed"])
    C, n = c["C"], c["n"]
    bound = float(2 ** (C - 1))
    bits = _bits(c["occ"], C, bound, c["p"], c["seed"])
    ebits = bits & np.packbits(rng.random(bits.size * 8) < 0.5, bitorder="little")
    o, d = _rays(n, c["seed"], bound, inside=(C > 1))
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    alive0 = np.sort(rng.choice(n, max(1, int(n * c["alive_frac"])), replace=False)).astype(np.int32)
    rng.shuffle(alive0)
    na, ns = alive0.size, c["n_step"]
    t0 = (nears + rng.random(n).astype(np.float32) * np.maximum(fars - nears, 0) * np.float32(0.7)).astype(np.float32)   # somewhere along the ray
    noises = rng.random(na).astype(np.float32) if c["perturb"] else np.zeros(na, np.float32)
    r0 = O.march_rays(na, ns, alive0, t0, o, d, bound, bits, C, 128, nears, fars, noises, align=128, dt_gamma=c["dtg"], max_steps=c["max_steps"],
                      edit_bitfield=ebits if c["edit"] else None)
    Mrows = r0[0].shape[0]
    xyzs = torch.zeros(Mrows, 3, device=DEV); dirs = torch.zeros(Mrows, 3, device=DEV); deltas = torch.zeros(Mrows, 2, device=DEV)
    alive, t = T(alive0), T(t0)
    if c["edit"]:
        eo = torch.zeros(Mrows, dtype=torch.bool, device=DEV)
        B.march_rays_distill(na, ns, alive, t, T(o), T(d), bound, c["dtg"], c["max_steps"], C, 128, T(bits), T(ebits), T(nears), T(fars), xyzs, dirs, deltas, eo, T(noises))
        assert np.array_equal(N(eo).astype(np.uint8), r0[3])
    else:
        B.march_rays(na, ns, alive, t, T(o), T(d), bound, c["dtg"], c["max_steps"], C, 128, T(bits), T(nears), T(fars), xyzs, dirs, deltas, T(noises))
    assert np.array_equal(N(xyzs), r0[0]) and np.array_equal(N(dirs), r0[1]) and np.array_equal(N(deltas), r0[2])
    sig = rng.exponential(20.0, Mrows).astype(np.float32); rgb = rng.uniform(0, 1, (Mrows, 3)).astype(np.float32)
    acc0 = [rng.uniform(0, 0.5, n).astype(np.float32) for _ in range(4)] + [rng.uniform(0, 0.5, (n, 3)).astype(np.float32)]
    acc = [T(a.copy()) for a in acc0]
    a0, tt0 = alive0.copy(), t0.copy()
    if c["edit"]:
        O.composite_rays(na, ns, a0, tt0, sig, rgb, r0[2], acc0[0], acc0[2], acc0[4], c["T_thresh"], weights_edit_sum=acc0[1], depth_edit=acc0[3], edit_occ=r0[3])
        rm.composite_rays_distill(na, ns, alive, t, T(sig), T(rgb), deltas, acc[0], acc[1], acc[2], acc[3], acc[4], eo, c["T_thresh"])
    else:
        O.composite_rays(na, ns, a0, tt0, sig, rgb, r0[2], acc0[0], acc0[2], acc0[4], c["T_thresh"])
        rm.composite_rays(na, ns, alive, t, T(sig), T(rgb), deltas, acc[0], acc[2], acc[4], c["T_thresh"])
    assert np.array_equal(N(alive), a0) and np.array_equal(N(t), tt0)
    for k, (a, b) in enumerate(zip(acc, acc0)):
        if c["edit"] or k not in (1, 3):
            assert np.allclose(N(a), b, atol=3e-5), k
    out, n_out = rm.compact_rays_alive(alive, na)
    keep = a0[a0 >= 0]
    assert int(n_out.item()) == keep.size and np.array_equal(N(out)[:keep.size], keep)


@settings(max_examples=100 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"degree": st.integers(1, 8), "B": st.integers(1, 3000), "unit": st.booleans(), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_sh_encode(c):
    """SH basis of degree 1-8 (shencoder.cu:27-439) and its input gradient, unit and non-unit directions (the polynomials are evaluated as
    they stand; the reference normalises nothing)"""
    O = _oracle()
    from laenerf_amd.shencoder import sh_encode
    rng = np.random.default_rng(c["seed"])
    d = rng.standard_normal((c["B"], 3)).astype(np.float32)
    if c["unit"]:
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)
    else:
        d *= np.float32(0.7)
    ref, ref_dd = O.sh_encode_forward(d, c["degree"], True)
    scale = 1.0 + float(np.abs(ref).max())
    di = T(d).requires_grad_()
    y = sh_encode(di, c["degree"], True)
    assert np.allclose(N(y), ref, rtol=2e-5, atol=2e-6 * scale)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.backward(T(g))
    assert np.allclose(N(di.grad), O.sh_encode_backward(g, ref_dd, c["degree"]), rtol=1e-4, atol=1e-4 * (1 + np.abs(ref_dd).max()))


def _close_f16(a, b, rel=4e-3, floor=2e-3):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.abs(a - b).max() <= rel * np.abs(b).max() + floor


@settings(max_examples=100 * FUZZ_SCALE, **FUZZ)
@given(st.fixed_dictionaries({"IN": st.sampled_from([16, 32, 48, 64]), "H": st.sampled_from([16, 32, 64, 64, 128]), "NL": st.integers(2, 4),
                              "tiles": st.integers(1, 90), "act": st.sampled_from([0, 0, 0, 3, 6]), "seed": st.integers(0, 2 ** 20)}))
def test_fuzz_ffmlp_forward_backward(c):
    """the fully fused MLP (ffmlp.cu:331-407, 410-518) over widths, depths, input sizes and batch sizes (multiples of 16: what the MFMA
    tiles own; the module pads): outputs, input gradient and weight gradient against the oracle, ReLU masks taken from the ORACLE's
    activations by feeding its forward buffer where the backward reads one"""
    O = _oracle()
    from laenerf_amd.backend import ffmlp_backend as F
    IN, H, NL, B, act = c["IN"], c["H"], c["NL"], 16 * c["tiles"], c["act"]
    rng = np.random.default_rng(c["seed"])
    nW = O.ffmlp_num_params(IN, H, NL)
    Wh = O.to_f16_bits(rng.uniform(-np.sqrt(3 / H), np.sqrt(3 / H), nW).astype(np.float32))
    Xh = O.to_f16_bits(rng.uniform(-1, 1, (B, IN)).astype(np.float32))
    ref_out, ref_fb = O.ffmlp_forward(Xh, Wh, IN, 16, H, NL, activation=act)
    out = torch.empty(B, 16, device=DEV, dtype=torch.half)
    F.ffmlp_inference(half_from_bits(Xh), half_from_bits(Wh), B, IN, 16, H, NL, act, 6, None, out)
    assert _close_f16(N(out), O.from_f16_bits(ref_out))
    if act != 0:
        return                                                 # the reference's backward knows ReLU only (ffmlp.cu:781: other activations are forward-only)
    Gh = O.to_f16_bits((rng.standard_normal((B, 16)) * 0.05).astype(np.float32))
    # The ReLU masks of the backward come from the FORWARD's activations.  The oracle's forward and the kernel's differ in the last
    # fp16 bit of a few activations (fp32 MFMA accumulate against the oracle's summation order), and a unit whose pre-activation lies
    # within that difference of zero (|z| ~ 1e-5: about one row in a thousand has one) is masked on one side and not on the other; that
    # row's input gradient then differs by the unit's whole contribution.  The deep run of this test (LAE_FUZZ_SCALE=20) found two such
    # draws (IN 64 / H 64 / 2 layers / 608 rows / seed 50; IN 48 / H 64 / 3 layers / 96 rows / seed 2515: ONE mask of 77 824 /
    # 18 432 each, tools/ffmlp_case_debug2.py).  In such a draw the oracle's backward is fed the KERNEL's forward buffer (the buffer-filling
    # mode 1 forward) -- what the reference's own backward reads too (ffmlp.py:35, 56); the recompute backward reproduced those masks.
    fbk = torch.empty(NL, B, H, device=DEV, dtype=torch.half); out1 = torch.empty_like(out)
    try:
        F.ffmlp_set_mode(1)
        F.ffmlp_forward(half_from_bits(Xh), half_from_bits(Wh), B, IN, 16, H, NL, 0, 6, fbk, out1)
    finally:
        F.ffmlp_set_mode(0)
    assert _close_f16(N(out1), O.from_f16_bits(ref_out)) and _close_f16(N(fbk), O.from_f16_bits(ref_fb))
    kb = bits_from_half(fbk)
    n_flip = int((((kb & 0x7fff) == 0) ^ ((ref_fb & 0x7fff) == 0)).sum())      # units masked on one side only (0 in all but ~1 draw in 300)
    ref_gw, ref_gi, _ = O.ffmlp_backward(Gh, Xh, Wh, kb if n_flip else ref_fb, IN, 16, H, NL, calc_grad_inputs=True)
    fused = F.fused_backward_available(IN, H, NL, 0)
    gi = torch.zeros(B, IN, device=DEV, dtype=torch.half); gw = torch.zeros(nW, device=DEV, dtype=torch.half)
    bb = None if fused else torch.empty(NL, B, H, device=DEV, dtype=torch.half)
    F.ffmlp_backward(half_from_bits(Gh), half_from_bits(Xh), half_from_bits(Wh), None if fused else half_from_bits(kb if n_flip else ref_fb), B, IN, 16, H, NL, 0, 6, True,
                     bb, gi, gw)
    assert _close_f16(N(gi), O.from_f16_bits(ref_gi), floor=5e-4)
    assert _close_f16(N(gw), O.from_f16_bits(ref_gw), rel=1e-2, floor=2e-3)
