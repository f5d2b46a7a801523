"""-m gpu: the repo's renderer driver on the HIP operators vs vectors captured from the reference's unmodified
nerf/renderer.py + nerf/network_ff.py (tests/golden/make_golden.py), plus full-size property checks."""
import numpy as np
import pytest
import torch

from conftest import golden
from gpu_util import DEV, N, T
from laenerf_amd.field import _nerf_field

pytestmark = pytest.mark.gpu


def build(g):
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    bound = int(g["bound"])
    net = NeRFNetwork(bound=bound, num_levels=16, log2_hashmap_size=10).to(DEV)
    assert tuple(net.encoder.embeddings.shape) == g["table"].shape and np.array_equal(N(net.encoder.offsets), g["offsets"])
    net.encoder.embeddings.data = T(g["table"])
    net.sigma_net.weights.data = T(g["sigma_w"])
    net.color_net.weights.data = T(g["color_w"])
    # the goldens ran the fp16-table path (make_golden.py::half_table_path) = what the fused field op computes; nothing is
    # patched here, and every test below asserts that the fused op (field._nerf_field) is what rendered
    assert net.fused_field
    r = NeRFRenderer(net, bound=bound, min_near=0.2).to(DEV)
    r.density_bitfield = T(g["bitfield"])
    return net, r


# tolerances of the e2e pins (round 2: depth 2e-3 / 3e-3, MLP gradients 5 %, table gradients 8 %, norm 5 %)
TOL = {"depth": 2e-5, "mlp_grad": 2e-3, "table_grad": 5e-3, "table_norm": 1e-3, "depth_distill": 1e-4}      # observed: 1.5e-7, 3e-4, 1.5e-3


@pytest.mark.parametrize("tag", ["b1", "b2"])
def test_train_render_and_gradients(tag):
    g = golden("e2e_" + tag)
    net, r = build(g)
    o, d = T(g["rays_o"]), T(g["rays_d"])
    net.train()
    calls0 = _nerf_field.calls
    with torch.autocast("cuda", dtype=torch.float16):
        res = r.render_train(o, d, bg_color=1, perturb=False, max_steps=256)
        loss = ((res["image"] - T(g["target"])) ** 2).mean()
    assert _nerf_field.calls == calls0 + 1                                     # the fused field op rendered, not the operator chain
    assert np.array_equal(N(r.step_counter[0]), g["train_counter"])            # sample count / ray count: exact
    # north_star tolerance: 1e-4 RGB at the golden run's settings (fp32 table, fp16 MLP weights)
    assert np.abs(N(res["image"]) - g["train_image"]).max() < 1e-4
    assert np.abs(N(res["weights_sum"]) - g["train_ws"]).max() < 1e-4
    hit = g["train_ws"] > 0
    assert np.abs(N(res["depth"])[hit] - g["train_depth"][hit]).max() < TOL["depth"]
    assert loss.item() == pytest.approx(float(g["train_loss"]), rel=1e-4)
    # fp16 gradients need the loss scale the reference trains with (GradScaler): unscaled, the table gradients of this
    # fixture are ~1e-6, i.e. a handful of fp16 subnormal steps.  Round 3: the golden ran the SAME precision path (fp16
    # table, half accumulate, half gradients, this loss scale: make_golden.py::half_table_path), so the gradients are
    # asserted at rounding level; what remains is the fused MLP's fp32 accumulate against the oracle FFMLP's and the order
    # in which the half gradient table is summed (exact sum rounded once here, one rounding per add there).
    SCALE = float(g["loss_scale"])
    (loss * SCALE).backward()
    dev = {}
    for name, ref in (("sigma_net", g["g_sigma_w"]), ("color_net", g["g_color_w"])):
        got = N(getattr(net, name).weights.grad) / SCALE
        dev[name] = np.abs(got - ref).max() / np.abs(ref).max()
        assert dev[name] < TOL["mlp_grad"], (name, dev)
    gt = N(net.encoder.embeddings.grad) / SCALE
    assert np.linalg.norm(gt) == pytest.approx(float(g["g_table_norm"]), rel=TOL["table_norm"])
    dev["table"] = np.abs(gt[::997] - g["g_table_sample"]).max() / np.abs(g["g_table_sample"]).max()
    assert dev["table"] < TOL["table_grad"], dev
    assert abs(int((gt != 0).sum()) - int(g["g_table_nonzero"])) <= 0.002 * int(g["g_table_nonzero"])      # the same entries are touched
    print("e2e deviations", tag, {k: float("%.3g" % v) for k, v in dev.items()},
          "depth", float(np.abs(N(res["depth"])[hit] - g["train_depth"][hit]).max()))
    # steady-state sizing with an under-estimated mean_count: overflowing rays drop to background
    r.mean_count = int(g["train2_mean_count"])
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        res2 = r.render_train(o, d, bg_color=1, perturb=False, max_steps=256)
    assert np.array_equal(N(res2["weights_sum"]) == 0, g["train2_ws"] == 0)
    assert np.abs(N(res2["image"]) - g["train2_image"]).max() < 1e-4


@pytest.mark.parametrize("tag", ["b1", "b2"])
def test_eval_and_distill_render(tag):
    g = golden("e2e_" + tag)
    net, r = build(g)
    o, d = T(g["rays_o"]), T(g["rays_d"])
    net.eval()
    with torch.autocast("cuda", dtype=torch.float16):
        for dc in (True, False, None):                                  # device-side compaction == host boolean mask == frame loop
            calls0 = _nerf_field.calls
            if dc is None:                                              # lae_render_frame: one backend call, no autograd op at all
                ev = r.render_eval(o, d, bg_color=1, max_steps=256, want_stats=True)
                assert _nerf_field.calls == calls0 and ev["stats"]["iterations"] > 0
            else:
                ev = r.render_eval(o, d, bg_color=1, max_steps=256, device_compaction=dc, frame_loop=False)
                assert _nerf_field.calls > calls0                       # every iteration of the operator loop used the fused field op
            assert np.abs(N(ev["image"]) - g["eval_image"]).max() < 1e-4
            hit = g["train_ws"] > 0
            assert np.abs(N(ev["depth"])[hit] - g["eval_depth"][hit]).max() < TOL["depth"]
        ds = r.render_distill(o, d, T(g["edit_bitfield"]), max_steps=256)
    for k, ref in (("image", "dist_image"), ("weights", "dist_weights"), ("weights_edit", "dist_weights_edit")):
        assert np.abs(N(ds[k]) - g[ref]).max() < 1e-4, k
    for k, ref in (("depth", "dist_depth"), ("depth_edit", "dist_depth_edit"), ("x_term", "dist_x_term")):
        assert np.abs(N(ds[k]) - g[ref]).max() < TOL["depth_distill"], k


def test_reference_backend_installation():
    """the HIP backend can be registered under the reference's extension names"""
    import sys
    from laenerf_amd import backend
    backend.install_as_reference_backends()
    import _raymarching, _gridencoder, _shencoder, _ffmlp      # noqa: F401
    for fn in ("near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
               "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "march_rays_distill",
               "composite_rays", "composite_rays_distill"):
        assert callable(getattr(_raymarching, fn))
    c = torch.randint(0, 128, (64, 3), dtype=torch.int32, device=DEV)
    idx = torch.empty(64, dtype=torch.int32, device=DEV)
    _raymarching.morton3D(c, 64, idx)
    back = torch.empty(64, 3, dtype=torch.int32, device=DEV)
    _raymarching.morton3D_invert(idx, 64, back)
    assert torch.equal(back, c)
    for m in ("_raymarching", "_gridencoder", "_shencoder", "_ffmlp"):
        del sys.modules[m]


def test_full_size_properties(O):
    """BASELINE cfg2 sizes (4096 rays, L=16, T=2^19): size-independent properties instead of a full oracle run"""
    from laenerf_amd import synthetic as S
    from laenerf_amd import raymarching as rm
    from laenerf_amd.gridencoder import GridEncoder
    o, d = S.lego_like_rays(4096, seed=0)
    bits = S.pack_bits_np(S.sphere_density_grid(), 10.0)
    to, td, tb = T(o), T(d), T(bits)
    aabb = T(np.array([-1, -1, -1, 1, 1, 1], np.float32))
    n, f = rm.near_far_from_aabb(to, td, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, 1.0, tb, 1, 128, n, f, counter, -1, True, 128, False, 0, 1024)
    r = N(rays); total = int(counter[0].item())
    assert r[:, 2].sum() == total and np.array_equal(r[:, 1], np.concatenate([[0], np.cumsum(r[:-1, 2])]))
    # every emitted sample lies in an occupied voxel (bit test on the host)
    p = N(xyzs)[:total].astype(np.float64)
    nidx = np.clip(0.5 * (p + 1) * 128, 0, 127).astype(np.int32)
    idx = O.morton3D(nidx).astype(np.int64)
    assert np.all((bits[idx >> 3] >> (idx & 7)) & 1)
    # a spot-check subset of rays against the oracle is not possible with device noise; determinism instead:
    counter.zero_()
    torch.manual_seed(5); a = rm.march_rays_train(to, td, 1.0, tb, 1, 128, n, f, counter, -1, True, 128, False, 0, 1024)
    counter.zero_()
    torch.manual_seed(5); b = rm.march_rays_train(to, td, 1.0, tb, 1, 128, n, f, counter, -1, True, 128, False, 0, 1024)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    # hash grid at full size: linearity in the table and a 4096-sample oracle spot check
    enc = GridEncoder(desired_resolution=2048).to(DEV)
    enc.embeddings.data.uniform_(-1, 1)
    pts = xyzs[:total]
    y1 = enc(pts, bound=1)
    enc.embeddings.data *= 2
    y2 = enc(pts, bound=1)
    assert torch.allclose(y2, 2 * y1, rtol=1e-6, atol=1e-6)
    sub = N(pts[:4096])
    ref, _ = O.grid_encode_forward((sub + 1) / 2, N(enc.embeddings), N(enc.offsets), enc.per_level_scale, 16, out_blc=True)
    assert np.allclose(N(y2[:4096]), ref, atol=1e-5)
    # backward at full size: sum of the table gradient == sum over samples of the output gradient (weights sum to 1)
    y = enc(pts, bound=1)
    g = torch.randn_like(y)
    y.backward(g)
    assert enc.embeddings.grad.double().sum().item() == pytest.approx(g.double().sum().item(), rel=1e-3, abs=1e-2)


def test_planned_grid_backward_equals_default():
    """march_train(plan_backward=True): the counting half of the table-gradient pass runs with the march (positions only);
    gradients must be identical to the default back-to-back pipeline on every level (exact fp16 sums, one rounding,
    split partitions merged in a fixed order)"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(3)
    net = NeRFNetwork(bound=1, log2_hashmap_size=19).to(DEV)
    net.encoder.embeddings.data.uniform_(-0.3, 0.3)
    r = NeRFRenderer(net, bound=1).to(DEV)
    r.density_bitfield = T(S.pack_bits_np(S.sphere_density_grid(), 10.0))
    o, d = S.lego_like_rays(1024, seed=4)
    o, d = T(o), T(d)
    net.train()
    grads = []
    for plan in (False, True):
        net.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            marched = r.march_train(o, d, perturb=False, plan_backward=plan)
            assert len(marched) == (7 if plan else 6)
            res = r.shade_train(marched, bg_color=1)
            loss = ((res["image"] - 0.3) ** 2).mean() * 1024.0
        loss.backward()
        grads.append([p.grad.clone() for p in (net.encoder.embeddings, net.sigma_net.weights, net.color_net.weights)])
    assert grads[0][0].abs().sum().item() > 0
    for a, b in zip(grads[0][1:], grads[1][1:]):
        assert torch.equal(a, b)                                            # MLP weights: fixed-order reductions
    assert torch.equal(grads[0][0], grads[1][0])


def test_fused_criterion_equals_separate_loss():
    """shade_train(gt=...): compositing forward + MSE + loss scale + compositing backward in ONE kernel ==
    composite_rays_train_blend followed by mse_loss_scaled and their backward kernels"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.losses import mse_loss_scaled
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(5)
    net = NeRFNetwork(bound=1, log2_hashmap_size=19).to(DEV)
    net.encoder.embeddings.data.uniform_(-0.3, 0.3)
    r = NeRFRenderer(net, bound=1).to(DEV)
    r.density_bitfield = T(S.pack_bits_np(S.sphere_density_grid(), 10.0))
    o, d = S.lego_like_rays(1000, seed=6)
    o, d = T(o), T(d)
    gt = torch.rand(1000, 3, device=DEV)
    scale = torch.tensor([512.0], device=DEV)
    net.train()
    out = []
    for fused in (False, True):
        net.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                res = r.render_train(o, d, bg_color=1, perturb=False, gt=gt, scaler=scale)
                loss = res["loss"]
            else:
                res = r.render_train(o, d, bg_color=1, perturb=False)
                loss = mse_loss_scaled(res["image"], gt, scale)
        (loss * 2.0).backward()                               # upstream gradient != 1 on purpose
        out.append((loss.detach().clone(), loss.unscaled.clone(), res["image"].detach().clone(),
                    [p.grad.clone() for p in (net.sigma_net.weights, net.color_net.weights, net.encoder.embeddings)]))
    (l0, u0, i0, g0), (l1, u1, i1, g1) = out
    # the fused node adds the squared errors workgroup by workgroup (fixed order), the separate criterion element by element:
    # the same sum to fp32 rounding; pixels and gradients are the same bits (an upstream gradient of 2 scales exactly)
    assert l0.item() == pytest.approx(l1.item(), rel=2e-6) and u0.item() == pytest.approx(u1.item(), rel=2e-6) and torch.equal(i0, i1)
    assert l0.item() == pytest.approx(512.0 * u0.item(), rel=1e-6) and l1.item() == pytest.approx(512.0 * u1.item(), rel=1e-6)
    assert torch.equal(g0[0], g1[0]) and torch.equal(g0[1], g1[1])
    assert torch.equal(g0[2], g1[2])


def test_deferred_loss_value_is_finished_by_the_backward_pass():
    """with a FusedAdam as the scaler the criterion's final one-block sum is taken off the step's critical path: the loss VALUE is
    NaN until the backward pass has run (the fused head's reduction launch carries it; FusedAdam.backward / step finish it
    otherwise) and then holds the same bits as the immediate sum; gradients are unaffected"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.optim import FusedAdam
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(5)
    net = NeRFNetwork(bound=1, log2_hashmap_size=14).to(DEV)
    net.encoder.embeddings.data.uniform_(-0.3, 0.3)
    r = NeRFRenderer(net, bound=1).to(DEV)
    r.density_bitfield = T(S.pack_bits_np(S.sphere_density_grid(), 10.0))
    opt = FusedAdam(net, param_groups=net.get_params(1e-2), init_scale=512.0)
    o, d = S.lego_like_rays(1000, seed=6)
    o, d = T(o), T(d)
    gt = torch.rand(1000, 3, device=DEV)
    net.train()
    vals = []
    for how in ("immediate", "head_backward", "optimizer_backward_without_head", "step_only"):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            marched = r.march_train(o, d, perturb=False)
            if how == "immediate":
                from laenerf_amd import raymarching as rm
                xyzs, dirs, deltas, rays, nears, fars = marched[:6]
                sigmas, rgbs = net(xyzs, dirs)
                loss, *_ = rm.composite_rays_train_blend_mse(sigmas, rgbs, deltas, rays, nears, fars, gt, 1, 1e-4, opt, defer_loss=False)
                assert torch.isfinite(loss).item() and torch.isfinite(loss.unscaled).item()
            else:
                loss = r.shade_train(marched, bg_color=1, gt=gt, scaler=opt)["loss"]
                assert torch.isnan(loss).item() and torch.isnan(loss.unscaled).item()      # not summed yet: loudly so
        if how in ("immediate", "head_backward"):
            opt.backward(loss)                                   # the fused head's backward carries the sum
        elif how == "optimizer_backward_without_head":
            from laenerf_amd import backend
            pend = dict(backend._pending_loss); backend._pending_loss.clear()          # nobody takes it along ...
            opt.backward(loss)
            assert torch.isnan(loss).item()
            backend._pending_loss.update(pend)
            opt.finish_loss()                                    # ... FusedAdam finishes it
        else:
            loss.backward()                                      # plain autograd backward: the head's backward carries it
        vals.append((loss.detach().clone(), loss.unscaled.clone(), net.sigma_net.shadow.grad_half.clone()))
        opt.zero_grad()
    for v in vals[1:]:
        assert torch.equal(v[0], vals[0][0]) and torch.equal(v[1], vals[0][1]) and torch.equal(v[2], vals[0][2])
    assert vals[0][0].item() == pytest.approx(512.0 * vals[0][1].item(), rel=1e-6)


def test_cfg0_run_path_train_step():
    """BASELINE configs[0]: lego 64x64, 1024 rays, L=4 hash grid, nn.Linear nets, cuda_ray off -> NeRFRenderer.run with
    num_steps 512 / upsample_steps 0 (main_nerf.py:32-35), fp32, one train step (forward, MSE, backward) against the
    reference's own NeRFNetwork + renderer.run executed by tests/golden/make_golden.py::gen_cfg0"""
    from laenerf_amd.network import NeRFNetworkLinear
    from laenerf_amd.renderer import NeRFRenderer
    g = golden("cfg0_run_step")
    net = NeRFNetworkLinear(bound=1, num_levels=4, log2_hashmap_size=14).to(DEV)
    assert np.array_equal(N(net.encoder.offsets), g["offsets"])
    net.encoder.embeddings.data = T(g["table"].astype(np.float32))
    for i, layer in enumerate(net.sigma_net):
        layer.weight.data = T(g[f"sigma_w{i}"])
    for i, layer in enumerate(net.color_net):
        layer.weight.data = T(g[f"color_w{i}"])
    r = NeRFRenderer(net, bound=1, min_near=0.2).to(DEV)
    net.train(); r.train()
    res = r.run(T(g["rays_o"])[None], T(g["rays_d"])[None], num_steps=int(g["num_steps"]), upsample_steps=0, bg_color=1, perturb=False)
    assert np.abs(N(res["image"][0]) - g["image"]).max() < 1e-4              # north_star RGB tolerance (observed ~1e-6)
    assert np.abs(N(res["weights_sum"]) - g["weights_sum"]).max() < 1e-4
    dep, dref = N(res["depth"][0]), g["depth"]
    assert np.array_equal(np.isnan(dep), np.isnan(dref)) and np.nanmax(np.abs(dep - dref)) < 1e-4
    loss = ((res["image"][0] - T(g["target"])) ** 2).mean()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    loss.backward()
    for i, layer in enumerate(net.sigma_net):
        ref = g[f"g_sigma_w{i}"]
        assert np.abs(N(layer.weight.grad) - ref).max() < 1e-3 * np.abs(ref).max() + 1e-9, f"sigma {i}"
    for i, layer in enumerate(net.color_net):
        ref = g[f"g_color_w{i}"]
        assert np.abs(N(layer.weight.grad) - ref).max() < 1e-3 * np.abs(ref).max() + 1e-9, f"color {i}"
    gt = N(net.encoder.embeddings.grad)
    assert np.linalg.norm(gt) == pytest.approx(float(g["g_table_norm"]), rel=1e-4)
    ref = g["g_table_sample"]
    assert np.abs(gt[::211] - ref).max() < 1e-3 * np.abs(ref).max() + 1e-9


def test_flower_shaped_full_size_properties(O):
    """BASELINE configs[2] sizes (scripts/configs_llff/flower.sh: bound 2 -> cascade 2, offset (0,0,1.5) -> cameras
    inside the box, min_near 0.2; 4096 rays, T=2^19, finest resolution 4096): size-independent properties + oracle spot
    checks on subsets the oracle finishes in seconds"""
    from laenerf_amd import synthetic as S
    from laenerf_amd import raymarching as rm
    from laenerf_amd.gridencoder import GridEncoder
    bound, C = 2.0, 2
    o, d = S.flower_like_rays(4096, seed=0)
    bits = S.pack_bits_np(S.flower_density_grid(), 10.0)
    assert bits.shape[0] == 2 * 128 ** 3 // 8                                  # 512 KiB bitfield
    to, td, tb = T(o), T(d), T(bits)
    aabb = T(np.array([-bound] * 3 + [bound] * 3, np.float32))
    n, f = rm.near_far_from_aabb(to, td, aabb, 0.2)
    n0, f0 = O.near_far_from_aabb(o, d, [-bound] * 3 + [bound] * 3, 0.2)
    assert np.array_equal(N(n), n0) and np.array_equal(N(f), f0) and (n0 == np.float32(0.2)).all()   # origins inside: near = min_near
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, bound, tb, C, 128, n, f, counter, -1, False, 128, False, 0, 1024)
    r = N(rays); total = int(counter[0].item())
    assert r[:, 2].sum() == total and np.array_equal(r[:, 1], np.concatenate([[0], np.cumsum(r[:-1, 2])]))
    # per-ray sample counts and positions of the first 384 rays == oracle (rays are independent; perturb off)
    sub = 384
    ref = O.march_rays_train(o[:sub], d[:sub], bound, bits, C, 128, n0[:sub], f0[:sub], np.zeros(sub, np.float32))
    assert np.array_equal(r[:sub, 2], ref[3][:, 2]) and np.array_equal(r[:sub, 1], ref[3][:, 1])
    m = int(ref[4][0])
    assert np.array_equal(N(xyzs)[:m], ref[0][:m]) and np.array_equal(N(deltas)[:m], ref[2][:m])
    # every sample lies in an occupied voxel of ITS cascade (level 1 where max|x| >= 1, raymarching.cu:42-47, 368)
    p = N(xyzs)[:total].astype(np.float64)
    level = (np.abs(p).max(1) >= 1.0).astype(np.int64)
    assert 0.2 < level.mean() < 0.8                                            # both cascades are exercised
    mip_bound = np.where(level == 1, 2.0, 1.0)[:, None]
    nidx = np.clip(0.5 * (p / mip_bound + 1) * 128, 0, 127).astype(np.int32)
    idx = level * 128 ** 3 + O.morton3D(nidx).astype(np.int64)
    assert np.all((bits[idx >> 3] >> (idx & 7)) & 1)
    # hash grid at flower size: desired_resolution 2048 * bound = 4096, T = 2^19 -> 6 328 848 entries
    enc = GridEncoder(desired_resolution=4096).to(DEV)
    assert int(enc.offsets[-1].item()) == 6328848
    enc.embeddings.data.uniform_(-1, 1)
    pts = xyzs[:total]
    y1 = enc(pts, bound=bound)
    subp = N(pts[:4096])
    refy, _ = O.grid_encode_forward((subp + bound) / (2 * bound), N(enc.embeddings), N(enc.offsets), enc.per_level_scale, 16, out_blc=True)
    assert np.allclose(N(y1[:4096]), refy, atol=1e-5)
    with torch.autocast("cuda", dtype=torch.float16):
        yh = enc(pts, bound=bound)
    th = O.to_f16_bits(N(enc.embeddings))
    refh, _ = O.grid_encode_forward((subp + bound) / (2 * bound), th, N(enc.offsets), enc.per_level_scale, 16, f16=True, out_blc=True)
    assert np.array_equal(N(yh[:4096]).astype(np.float16).view(np.uint16), refh)          # fp16 table path: bit exact
    y = enc(pts, bound=bound)
    gr = torch.randn_like(y)
    y.backward(gr)
    assert enc.embeddings.grad.double().sum().item() == pytest.approx(gr.double().sum().item(), rel=1e-3, abs=1e-2)


def test_deferred_loss_is_keyed_by_the_criterion_node_not_by_the_process():
    """LAENeRF's flow holds two models, two optimizers and one scaler in one process (nerf/utils.py:969-972, 1041-1043).  The
    deferred loss value of a criterion node is taken along ONLY by the head backward that consumes that node's gradient, on the
    stream it was made on (round 4: one slot per process -- whichever head backward ran next summed whatever was pending).
    Interleaved: two NeRF models (A forward, B forward, B backward, A backward), a LAENeRF palette step between A's forward and
    backward, and backward on another stream than forward; every loss value holds the bits of the immediate sum."""
    from types import SimpleNamespace
    from laenerf_amd import backend, synthetic as S
    from laenerf_amd.editing import LAENeRF
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.optim import FusedAdam
    from laenerf_amd.renderer import NeRFRenderer

    def model(seed):
        torch.manual_seed(seed)
        net = NeRFNetwork(bound=1, log2_hashmap_size=14).to(DEV)
        net.encoder.embeddings.data.uniform_(-0.3, 0.3)
        r = NeRFRenderer(net, bound=1).to(DEV)
        r.density_bitfield = T(S.pack_bits_np(S.sphere_density_grid(), 10.0))
        net.train()
        return net, r, FusedAdam(net, param_groups=net.get_params(1e-2), init_scale=512.0)
    (netA, rA, optA), (netB, rB, optB) = model(5), model(6)
    oA, dA = (T(x) for x in S.lego_like_rays(1000, seed=6))
    oB, dB = (T(x) for x in S.lego_like_rays(1400, seed=7))
    gtA, gtB = torch.rand(1000, 3, device=DEV), torch.rand(1400, 3, device=DEV)

    def fwd(r, opt, o, d, gt, defer):
        from laenerf_amd import raymarching as rm
        with torch.autocast("cuda", dtype=torch.float16):
            marched = r.march_train(o, d, perturb=False)
            if defer:
                return r.shade_train(marched, bg_color=1, gt=gt, scaler=opt)["loss"]
            xyzs, dirs, deltas, rays, nears, fars = marched[:6]
            sigmas, rgbs = r.network(xyzs, dirs) if hasattr(r, "network") else opt.module(xyzs, dirs)
            return rm.composite_rays_train_blend_mse(sigmas, rgbs, deltas, rays, nears, fars, gt, 1, 1e-4, opt, defer_loss=False)[0]

    def immediate(net, r, opt, o, d, gt):
        from laenerf_amd import raymarching as rm
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            xyzs, dirs, deltas, rays, nears, fars = r.march_train(o, d, perturb=False)[:6]
            sigmas, rgbs = net(xyzs, dirs)
            loss = rm.composite_rays_train_blend_mse(sigmas, rgbs, deltas, rays, nears, fars, gt, 1, 1e-4, opt, defer_loss=False)[0]
        opt.backward(loss)
        out = (loss.detach().clone(), loss.unscaled.clone(), net.sigma_net.shadow.grad_half.clone())
        opt.zero_grad()
        return out
    refA, refB = immediate(netA, rA, optA, oA, dA, gtA), immediate(netB, rB, optB, oB, dB, gtB)

    def same(loss, net, ref):
        assert torch.equal(loss.detach(), ref[0]) and torch.equal(loss.unscaled, ref[1]) and torch.equal(net.sigma_net.shadow.grad_half, ref[2])

    # 1. two NeRF models interleaved on one stream: each value rides in ITS OWN head backward
    backend.flush_pending_loss()
    st0 = dict(backend.deferred_loss_stats)
    lA = fwd(rA, optA, oA, dA, gtA, True)
    lB = fwd(rB, optB, oB, dB, gtB, True)
    assert torch.isnan(lA).item() and torch.isnan(lB).item() and len(backend._pending_loss) == 2
    lB.backward(gradient=torch.ones_like(lB))                  # plain autograd backward of B: must not touch A's value
    assert torch.isnan(lA).item() and len(backend._pending_loss) == 1
    same(lB, netB, refB)
    optA.backward(lA)
    same(lA, netA, refA)
    st1 = dict(backend.deferred_loss_stats)
    assert st1["carried"] - st0["carried"] == 2 and st1["flushed"] == st0["flushed"] and not backend._pending_loss
    optA.zero_grad(); optB.zero_grad()

    # 2. a LAENeRF palette step (its own FusedAdam) between A's forward and A's backward
    params = SimpleNamespace(bound=1, num_palette_bases=8, style_weight=0, weight_loss_uniform=1e-3, weight_loss_non_uniform=1e-3,
                             offset_loss=1e-2, palette_loss_valid=1.0, palette_loss_distinct=1e-2)
    torch.manual_seed(7)
    m = LAENeRF(params, dir_encoding="sphere_harmonics").to(DEV).train()
    optS = FusedAdam(m, param_groups=m.get_params(1e-3), betas=(0.9, 0.999), eps=1e-8)
    x = (torch.rand(5120, 3, device=DEV) - 0.5) * 0.6
    dd = torch.nn.functional.normalize(torch.randn(5120, 3, device=DEV), dim=-1)
    tgt = torch.rand(5120, 3, device=DEV)
    lA = fwd(rA, optA, oA, dA, gtA, True)
    with torch.autocast("cuda", dtype=torch.float16):
        lS = m.forward_train_loss(x, dd, tgt, params, optS, with_palet_loss=True)[0]
    optS.backward(lS)                                          # FusedAdam.backward finishes what is pending: A's value too, on A's stream
    optS.step()
    assert torch.isfinite(lS).item()
    optA.backward(lA)
    same(lA, netA, refA)
    optA.zero_grad()

    # 3. forward under one stream context, backward called under another: autograd runs every node's backward on the stream of
    # its forward, so the value is still carried, on s1
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        lA = fwd(rA, optA, oA, dA, gtA, True)
    s2.wait_stream(s1)
    st2 = dict(backend.deferred_loss_stats)
    with torch.cuda.stream(s2):
        lA.backward(gradient=torch.ones_like(lA))
    assert backend.deferred_loss_stats["carried"] == st2["carried"] + 1 and not backend._pending_loss
    torch.cuda.synchronize()
    same(lA, netA, refA)
    optA.zero_grad()

    # 4. an entry whose partials were written on ANOTHER stream than the one the head backward runs on is not taken along (no
    # ordering against them there): it is finished on its own stream by flush_pending_loss
    lA = fwd(rA, optA, oA, dA, gtA, True)
    (key, e), = backend._pending_loss.items()
    s2.wait_stream(torch.cuda.current_stream())
    backend._pending_loss[key] = e[:5] + (s2.cuda_stream,) + e[6:]
    st3 = dict(backend.deferred_loss_stats)
    lA.backward(gradient=torch.ones_like(lA))
    assert backend.deferred_loss_stats["carried"] == st3["carried"] and len(backend._pending_loss) == 1 and torch.isnan(lA).item()
    optA.finish_loss()
    assert backend.deferred_loss_stats["flushed"] == st3["flushed"] + 1 and not backend._pending_loss
    torch.cuda.synchronize()
    same(lA, netA, refA)
    optA.zero_grad()
