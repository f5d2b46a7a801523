"""-m gpu: raymarching HIP kernels (through the C ABI) vs the CPU oracle. Integer outputs, t-sequences and
sample positions are bit-exact; composited floats within 1e-5 (fast-exp vs libm expf)."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, N, T, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rm():
    from laenerf_amd import raymarching
    return raymarching


def test_near_far_bit_exact(O, rm):
    rng = np.random.default_rng(0)
    o = rng.uniform(-3, 3, (4096, 3)).astype(np.float32)
    d = rng.standard_normal((4096, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:16, 0] = 0; d[16:32, 1] = 0; d[32:40] = [0, 0, 1]            # zero components -> 1/0 = inf slabs
    o[40:80] *= 0.2                                                   # origins inside the box
    for aabb, mn in (([-1, -1, -1, 1, 1, 1], 0.2), ([-2, -1, -0.5, 2, 1.5, 2], 0.05)):
        n, f = rm.near_far_from_aabb(T(o), T(d), T(np.array(aabb, np.float32)), mn)
        n0, f0 = O.near_far_from_aabb(o, d, aabb, mn)
        assert np.array_equal(N(n), n0) and np.array_equal(N(f), f0)


def test_morton_packbits_sph_bit_exact(O, rm):
    rng = np.random.default_rng(1)
    c = rng.integers(0, 128, (100000, 3)).astype(np.int32)
    m = rm.morton3D(T(c))
    assert np.array_equal(N(m), O.morton3D(c))
    assert np.array_equal(N(rm.morton3D_invert(m)), c)
    g = rng.uniform(-1, 30, (2, 128 ** 3 // 64)).astype(np.float32); g[0, ::5] = 10.0
    assert np.array_equal(N(rm.packbits(T(g), 10.0)), O.packbits(g, 10.0))
    o = rng.uniform(-0.5, 0.5, (1000, 3)).astype(np.float32)
    d = rng.standard_normal((1000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    assert np.allclose(N(rm.sph_from_ray(T(o), T(d), 3.0)), O.sph_from_ray(o, d, 3.0), atol=2e-6)


@pytest.mark.parametrize("C,bound,dtg,max_steps", [(1, 1.0, 0.0, 1024), (2, 2.0, 0.0, 1024), (2, 2.0, 1 / 128, 512),
                                                   (1, 1.0, 1 / 64, 64), (3, 4.0, 0.0, 256)])
def test_march_rays_train_bit_exact(O, C, bound, dtg, max_steps):
    from laenerf_amd.backend import raymarching_backend as B
    sc = scene(C, bound, n_rays=777, seed=C)
    Nr = 777
    noises = np.random.default_rng(5).random(Nr).astype(np.float32)
    ref = O.march_rays_train(sc["o"], sc["d"], bound, sc["bits"], C, 128, sc["nears"], sc["fars"], noises,
                             dt_gamma=dtg, max_steps=max_steps)
    total = int(ref[4][0])
    for M in (Nr * max_steps, total, max(total // 2, 1)):             # exact fit and overflow-drop
        ref = O.march_rays_train(sc["o"], sc["d"], bound, sc["bits"], C, 128, sc["nears"], sc["fars"], noises, M=M,
                                 dt_gamma=dtg, max_steps=max_steps)
        # poisoned buffers: the kernel itself must zero the rows no ray owns (the reference relies on torch.zeros)
        xyzs = torch.full((M, 3), float("nan"), device=DEV); dirs = torch.full((M, 3), float("nan"), device=DEV)
        deltas = torch.full((M, 2), float("nan"), device=DEV)
        rays = torch.empty(Nr, 3, dtype=torch.int32, device=DEV); counter = torch.zeros(2, dtype=torch.int32, device=DEV)
        rows_end = torch.full((1,), -1, dtype=torch.int32, device=DEV)
        B.march_rays_train(T(sc["o"]), T(sc["d"]), T(sc["bits"]), bound, dtg, max_steps, Nr, C, 128, M, T(sc["nears"]),
                           T(sc["fars"]), xyzs, dirs, deltas, rays, counter, T(noises), rows_end)
        assert np.array_equal(N(counter), ref[4])
        fit = ref[3][(ref[3][:, 2] > 0) & (ref[3][:, 1] + ref[3][:, 2] <= M)]
        assert int(rows_end.item()) == (int((fit[:, 1] + fit[:, 2]).max()) if len(fit) else 0)
        assert np.array_equal(N(rays), ref[3])                       # ids, offsets, counts: bit exact
        assert np.array_equal(N(xyzs), ref[0]) and np.array_equal(N(dirs), ref[1]) and np.array_equal(N(deltas), ref[2])


@pytest.mark.parametrize("C,bound,dtg,perturb", [(C, b, g, p) for (C, b) in ((1, 1.0), (2, 2.0), (4, 8.0))
                                                for g in (0.0, 1 / 128) for p in (False, True)])
def test_march_integer_outputs_match_both_oracle_flavours(O, C, bound, dtg, perturb):
    """`rays` / `counter` of the HIP marcher equal the oracle built with fused AND with un-fused multiply-adds
    (tests/test_oracle_flavours_cpu.py): the bit-exactness of indices / offsets does not rest on the nvcc
    contraction model; positions agree with the un-fused build to 1 ulp of the larger operand."""
    from laenerf_amd.backend import raymarching_backend as B
    Nr = 1024
    sc = scene(C, bound, n_rays=Nr, seed=10 + C)
    noises = np.random.default_rng(3).random(Nr).astype(np.float32) if perturb else np.zeros(Nr, np.float32)
    M = Nr * 1024
    xyzs = torch.empty(M, 3, device=DEV); dirs = torch.empty(M, 3, device=DEV); deltas = torch.empty(M, 2, device=DEV)
    rays = torch.empty(Nr, 3, dtype=torch.int32, device=DEV); counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    B.march_rays_train(T(sc["o"]), T(sc["d"]), T(sc["bits"]), bound, dtg, 1024, Nr, C, 128, M, T(sc["nears"]), T(sc["fars"]),
                       xyzs, dirs, deltas, rays, counter, T(noises))
    for fl in ("fma", "nofma"):
        with O.flavour(fl):
            ref = O.march_rays_train(sc["o"], sc["d"], bound, sc["bits"], C, 128, sc["nears"], sc["fars"], noises,
                                     dt_gamma=dtg, max_steps=1024)
        assert np.array_equal(N(rays), ref[3]) and np.array_equal(N(counter), ref[4]), fl
        total = int(ref[4][0])
        if fl == "fma":
            assert np.array_equal(N(xyzs)[:total], ref[0][:total]) and np.array_equal(N(deltas)[:total], ref[2][:total])
        else:
            assert np.abs(N(xyzs)[:total] - ref[0][:total]).max() <= 4.8e-7


def test_march_rays_train_record_overflow_falls_back_to_walk(O):
    """a half-occupied random grid at bound 8 spreads a ray's 1024 samples over more candidate chunks than the emit
    pass keeps records for (24): those rays are walked again and must still be bit-identical"""
    from laenerf_amd.backend import raymarching_backend as B
    C, bound, Nr, max_steps = 4, 8.0, 333, 1024
    sc = scene(C, bound, n_rays=Nr, seed=7, radius_cam=6.0)
    bits = np.random.default_rng(11).integers(0, 256, sc["bits"].shape[0]).astype(np.uint8)
    noises = np.random.default_rng(6).random(Nr).astype(np.float32)
    M = Nr * max_steps
    ref = O.march_rays_train(sc["o"], sc["d"], bound, bits, C, 128, sc["nears"], sc["fars"], noises, M=M, max_steps=max_steps)
    assert (ref[3][:, 2] == max_steps).sum() > 50                    # many rays run into the cap, i.e. > 2000 candidates
    xyzs = torch.empty(M, 3, device=DEV); dirs = torch.empty(M, 3, device=DEV); deltas = torch.empty(M, 2, device=DEV)
    rays = torch.empty(Nr, 3, dtype=torch.int32, device=DEV); counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    B.march_rays_train(T(sc["o"]), T(sc["d"]), T(bits), bound, 0.0, max_steps, Nr, C, 128, M, T(sc["nears"]), T(sc["fars"]),
                       xyzs, dirs, deltas, rays, counter, T(noises))
    assert np.array_equal(N(rays), ref[3]) and np.array_equal(N(counter), ref[4])
    assert np.array_equal(N(xyzs), ref[0]) and np.array_equal(N(dirs), ref[1]) and np.array_equal(N(deltas), ref[2])


def test_march_rays_train_wrapper_modes(O, rm):
    """the autograd.Function front-end: trimming / mean_count sizing (raymarching.py:196-231)"""
    sc = scene(1, 1.0, n_rays=300, seed=2)
    o, d, bits, n, f = T(sc["o"]), T(sc["d"]), T(sc["bits"]), T(sc["nears"]), T(sc["fars"])
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, bits, 1, 128, n, f, counter, -1, False, 128, False, 0, 1024)
    total = int(counter[0].item())
    assert xyzs.shape[0] == total + (128 - total % 128) and counter[1].item() == 300
    ref = O.march_rays_train(sc["o"], sc["d"], 1.0, sc["bits"], 1, 128, sc["nears"], sc["fars"], np.zeros(300))
    assert np.array_equal(N(xyzs)[:total], ref[0][:total]) and np.all(N(deltas)[total:] == 0)
    counter.zero_()
    x2, _, _, r2 = rm.march_rays_train(o, d, 1.0, bits, 1, 128, n, f, counter, 1000, False, 128, False, 0, 1024)
    assert x2.shape[0] == 1024 and np.array_equal(N(r2), ref[3])
    # perturb=True draws noises on the device: counts change but invariants hold
    counter.zero_()
    x3, _, dl3, r3 = rm.march_rays_train(o, d, 1.0, bits, 1, 128, n, f, counter, -1, True, 128, False, 0, 1024)
    r3 = N(r3)
    assert r3[:, 2].sum() == counter[0].item() and np.array_equal(r3[:, 1], np.concatenate([[0], np.cumsum(r3[:-1, 2])]))
    # empty / ragged: zero rays, and rays that all miss
    e = torch.zeros(0, 3, device=DEV)
    xe, _, _, re_ = rm.march_rays_train(e, e, 1.0, bits, 1, 128, torch.zeros(0, device=DEV), torch.zeros(0, device=DEV), None, -1, False, 128, False, 0, 64)
    assert re_.shape == (0, 3) and xe.shape[0] == 0          # zeros(0,3)[:128] is empty, as in the reference
    far_o = T(np.full((64, 3), 9.0, np.float32)); up = T(np.tile(np.array([[0, 1, 0]], np.float32), (64, 1)))
    nn_, ff_ = rm.near_far_from_aabb(far_o, up, T(np.array([-1, -1, -1, 1, 1, 1], np.float32)), 0.2)
    c = torch.zeros(2, dtype=torch.int32, device=DEV)
    xm, _, _, rmiss = rm.march_rays_train(far_o, up, 1.0, bits, 1, 128, nn_, ff_, c, -1, False, 128, False, 0, 64)
    assert c[0].item() == 0 and np.all(N(rmiss)[:, 2] == 0)


def test_composite_train_forward_backward(O, rm):
    sc = scene(1, 1.0, n_rays=600, seed=3)
    x, dd, dl, rays, cnt = O.march_rays_train(sc["o"], sc["d"], 1.0, sc["bits"], 1, 128, sc["nears"], sc["fars"], np.zeros(600))
    M = int(cnt[0]) + 128
    rng = np.random.default_rng(9)
    sig = rng.uniform(0, 50, M).astype(np.float32); rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    rays_shuf = rays[rng.permutation(600)]                         # row order must not matter
    for Mlim, rr in ((M, rays), (M, rays_shuf), (M // 2, rays)):
        ws0, dep0, img0 = O.composite_rays_train_forward(sig[:Mlim], rgb[:Mlim], dl[:Mlim], rr, 1e-4)
        s, c = T(sig[:Mlim]).requires_grad_(), T(rgb[:Mlim]).requires_grad_()
        ws, dep, img = rm.composite_rays_train(s, c, T(dl[:Mlim]), T(rr), 1e-4)
        assert np.allclose(N(ws), ws0, atol=2e-6) and np.allclose(N(img), img0, atol=2e-6) and np.allclose(N(dep), dep0, atol=2e-5)
        gws = rng.standard_normal(600).astype(np.float32); gimg = rng.standard_normal((600, 3)).astype(np.float32)
        torch.autograd.backward([ws, img], [T(gws), T(gimg)])
        gs0, gc0 = O.composite_rays_train_backward(gws, gimg, sig[:Mlim], rgb[:Mlim], dl[:Mlim], rr, ws0, img0, 1e-4)
        assert np.allclose(N(s.grad), gs0, atol=3e-5, rtol=1e-4) and np.allclose(N(c.grad), gc0, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("C,bound,edit", [(1, 1.0, False), (2, 2.0, False), (2, 2.0, True)])
def test_inference_loop_trace(O, rm, C, bound, edit):
    """march_rays(+distill) / composite_rays(+distill) / compaction, iteration by iteration vs the oracle"""
    sc = scene(C, bound, n_rays=500, seed=7)
    Nr = 500
    o, d, bits, n, f = T(sc["o"]), T(sc["d"]), T(sc["bits"]), T(sc["nears"]), T(sc["fars"])
    eg = sc["grid"].copy(); eg[:, ::3] = 0
    from laenerf_amd import synthetic as S
    ebits = S.pack_bits_np(eg, 10.0)
    field = lambda p: (30 * np.exp(-3 * (p ** 2).sum(-1)).astype(np.float32), (0.5 + 0.5 * np.cos(2 * p)).astype(np.float32))
    acc0 = [np.zeros(Nr, np.float32) for _ in range(4)] + [np.zeros((Nr, 3), np.float32)]
    acc = [torch.zeros(Nr, device=DEV) for _ in range(4)] + [torch.zeros(Nr, 3, device=DEV)]
    alive0, t0 = np.arange(Nr, dtype=np.int32), sc["nears"].copy()
    alive, t = torch.arange(Nr, dtype=torch.int32, device=DEV), n.clone()
    step = 0
    while step < 1024 and alive0.size:
        na = alive0.size
        ns = max(min(Nr // na, 8), 1)
        noises = (np.random.default_rng(step).random(na).astype(np.float32) if step == 0 else np.zeros(na, np.float32))
        r0 = O.march_rays(na, ns, alive0, t0, sc["o"], sc["d"], bound, sc["bits"], C, 128, sc["nears"], sc["fars"], noises, align=128,
                          edit_bitfield=ebits if edit else None)
        from laenerf_amd.backend import raymarching_backend as B
        Mrows = r0[0].shape[0]
        xyzs = torch.zeros(Mrows, 3, device=DEV); dirs = torch.zeros(Mrows, 3, device=DEV); deltas = torch.zeros(Mrows, 2, device=DEV)
        if edit:
            eo = torch.zeros(Mrows, dtype=torch.bool, device=DEV)
            B.march_rays_distill(na, ns, alive, t, o, d, bound, 0.0, 1024, C, 128, bits, T(ebits), n, f, xyzs, dirs, deltas, eo, T(noises))
            assert np.array_equal(N(eo).astype(np.uint8), r0[3])
        else:
            B.march_rays(na, ns, alive, t, o, d, bound, 0.0, 1024, C, 128, bits, n, f, xyzs, dirs, deltas, T(noises))
        assert np.array_equal(N(xyzs), r0[0]) and np.array_equal(N(dirs), r0[1]) and np.array_equal(N(deltas), r0[2])
        s0, c0 = field(r0[0])
        if edit:
            O.composite_rays(na, ns, alive0, t0, s0, c0, r0[2], acc0[0], acc0[2], acc0[4], 1e-4, weights_edit_sum=acc0[1],
                             depth_edit=acc0[3], edit_occ=r0[3])
            rm.composite_rays_distill(na, ns, alive, t, T(s0), T(c0), deltas, acc[0], acc[1], acc[2], acc[3], acc[4], eo, 1e-4)
        else:
            O.composite_rays(na, ns, alive0, t0, s0, c0, r0[2], acc0[0], acc0[2], acc0[4], 1e-4)
            rm.composite_rays(na, ns, alive, t, T(s0), T(c0), deltas, acc[0], acc[2], acc[4], 1e-4)
        assert np.array_equal(N(alive)[:na], alive0)                         # -1 marks bit exact
        assert np.array_equal(N(t), t0)
        alive0 = np.ascontiguousarray(alive0[alive0 >= 0])
        out, n_out = rm.compact_rays_alive(alive, na)
        assert n_out.item() == alive0.size and np.array_equal(N(out)[:alive0.size], alive0)
        alive = out[:alive0.size].contiguous()
        step += ns
    for a, a0 in zip(acc, acc0):
        assert np.allclose(N(a), a0, atol=3e-5)


def test_compaction_sizes(rm):
    rng = np.random.default_rng(3)
    for n in (1, 63, 64, 65, 255, 256, 257, 1000, 300000):
        v = rng.integers(-1, 50, n).astype(np.int32); v[v < 10] = -1
        out, cnt = rm.compact_rays_alive(T(v))
        keep = v[v >= 0]
        assert cnt.item() == keep.size and np.array_equal(N(out)[:keep.size], keep)


@pytest.mark.parametrize("bg_mode", ["const", "rgb", "per_ray"])
def test_composite_blend_equals_separate_ops(rm, bg_mode):
    """composite + bg blend + depth normalisation in one kernel == renderer.py:318-325 built from separate ops;
    its backward defines every gradient row (no pre-zeroing) and folds the blend's gradient in"""
    from laenerf_amd.backend import raymarching_backend as B
    sc = scene(1, 1.0, n_rays=500, seed=4)
    o, d, bits, n, f = T(sc["o"]), T(sc["d"]), T(sc["bits"]), T(sc["nears"]), T(sc["fars"])
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for mean_count in (-1, 20000):                                   # trimmed buffers / fixed-size buffers with overflow drop
        counter.zero_()
        xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, bits, 1, 128, n, f, counter, mean_count, False, 128, False, 0, 1024)
        M = xyzs.shape[0]
        sig = (torch.rand(M, device=DEV, generator=gen) * 60).requires_grad_()        # dense enough for early stops
        rgb = torch.rand(M, 3, device=DEV, generator=gen).requires_grad_()
        bg = {"const": 1, "rgb": torch.tensor([0.2, 0.5, 0.9], device=DEV), "per_ray": torch.rand(500, 3, device=DEV, generator=gen)}[bg_mode]
        ws, dep, img = rm.composite_rays_train_blend(sig, rgb, deltas, rays, n, f, bg, 1e-4)
        gimg = torch.randn(500, 3, device=DEV, generator=gen); gws = torch.randn(500, device=DEV, generator=gen)
        torch.autograd.backward([img, ws], [gimg, gws])
        g1s, g1c = sig.grad.clone(), rgb.grad.clone(); sig.grad = None; rgb.grad = None
        ws0, dep0, img0 = rm.composite_rays_train(sig, rgb, deltas, rays, 1e-4)
        img_b = img0 + (1 - ws0).unsqueeze(-1) * bg
        dep_b = torch.clamp(dep0 - n, min=0) / (f - n)
        torch.autograd.backward([img_b, ws0], [gimg, gws])
        assert torch.equal(ws, ws0) and torch.equal(img, img_b)
        assert torch.equal(torch.nan_to_num(dep, nan=-1.0), torch.nan_to_num(dep_b, nan=-1.0))
        assert torch.allclose(g1s, sig.grad, rtol=1e-5, atol=1e-6) and torch.equal(g1c, rgb.grad)
        # coverage: the backward kernel must overwrite every row of poisoned gradient buffers
        gs = torch.full_like(sig, float("nan")); gc = torch.full_like(rgb, float("nan"))
        bgr = bg.contiguous() if bg_mode == "per_ray" else None
        bgc = (1.0, 1.0, 1.0) if bg_mode == "const" else (0.2, 0.5, 0.9)
        B.composite_rays_train_backward_blend(gws, gimg, sig.detach(), rgb.detach(), deltas, rays, ws0.detach(), img0.detach(), M, 500,
                                              1e-4, bgr, bgc, rays.rows_end, gs, gc)
        assert torch.isfinite(gs).all() and torch.isfinite(gc).all()
        assert torch.allclose(gs, g1s, rtol=1e-5, atol=1e-6) and torch.equal(gc, g1c)


@pytest.mark.parametrize("bg_mode", ["const", "per_ray"])
def test_composite_train_step_equals_the_three_kernels(rm, bg_mode):
    """lae_composite_rays_train_step (compositing forward + blend, MSE criterion, compositing backward in one launch) against
    lae_composite_rays_train_forward_blend + lae_mse_loss_forward + lae_composite_rays_train_backward_blend on the same
    buffers: rays of several 64-sample passes, early stops (dense sigmas), rays without samples, a trimmed and a fixed-size
    sample buffer that drops overflowing rays, poisoned gradient buffers (every row must be written).  Pixels and
    gradients: the same bits; the loss: the same sum in another order."""
    from laenerf_amd import _lib
    from laenerf_amd.backend import raymarching_backend as B
    N_ = 700
    sc = scene(1, 1.0, n_rays=N_, seed=9)
    o, d, bits, n, f = T(sc["o"]), T(sc["d"]), T(sc["bits"]), T(sc["nears"]), T(sc["fars"])
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    scale = torch.tensor([1024.0], device=DEV)
    for mean_count, dens in ((-1, 60.0), (-1, 0.5), (30000, 5.0)):
        counter.zero_()
        xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, bits, 1, 128, n, f, counter, mean_count, False, 128, False, 0, 1024)
        M = xyzs.shape[0]
        assert int(N(rays)[:, 2].max()) > 128 and int((N(rays)[:, 2] == 0).sum()) > 0       # multi-pass rays and empty rays
        sig = torch.rand(M, device=DEV, generator=gen) * dens
        rgb = torch.rand(M, 3, device=DEV, generator=gen)
        target = torch.rand(N_, 3, device=DEV, generator=gen)
        bgr = torch.rand(N_, 3, device=DEV, generator=gen) if bg_mode == "per_ray" else None
        bgc = (1.0, 1.0, 1.0)
        new = lambda *shape: torch.full(shape, float("nan"), device=DEV)
        # the three kernels
        ws0, dp0, im0, do0, io0 = new(N_), new(N_), new(N_, 3), new(N_), new(N_, 3)
        B.composite_rays_train_forward_blend(sig, rgb, deltas, rays, M, N_, 1e-4, n, f, bgr, bgc, ws0, dp0, im0, do0, io0)
        loss0, gi0 = new(2), new(N_, 3)
        _lib.check(_lib.load().lae_mse_loss_forward(io0.data_ptr(), target.data_ptr(), io0.numel(), scale.data_ptr(), loss0.data_ptr(),
                                                    gi0.data_ptr(), _lib.stream()), "mse")
        gs0, gc0 = new(M), new(M, 3)
        B.composite_rays_train_backward_blend(None, gi0, sig, rgb, deltas, rays, ws0, im0, M, N_, 1e-4, bgr, bgc, rays.rows_end, gs0, gc0)
        # the one kernel
        ws1, dp1, im1, do1, io1, gi1, gs1, gc1, loss1 = new(N_), new(N_), new(N_, 3), new(N_), new(N_, 3), new(N_, 3), new(M), new(M, 3), new(2)
        part = new((N_ + 3) // 4)
        B.composite_rays_train_step(sig, rgb, deltas, rays, M, N_, 1e-4, n, f, bgr, bgc, rays.rows_end, target, scale, ws1, dp1, im1, do1,
                                    io1, gi1, gs1, gc1, loss1, part)
        for a_, b_ in ((ws0, ws1), (dp0, dp1), (im0, im1), (io0, io1), (gi0, gi1), (gs0, gs1), (gc0, gc1)):
            assert torch.isfinite(b_).all() and torch.equal(a_, b_)
        assert torch.equal(torch.nan_to_num(do0, nan=-1.0), torch.nan_to_num(do1, nan=-1.0))
        assert loss1[1].item() == pytest.approx(loss0[1].item(), rel=2e-6) and loss1[0].item() == pytest.approx(1024.0 * loss1[1].item(), rel=1e-6)
        ref = torch.nn.functional.mse_loss(io1, target).item()
        assert loss1[1].item() == pytest.approx(ref, rel=1e-5)
