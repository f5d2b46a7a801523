"""-m gpu: FusedAdam (csrc/optimizer.hip) against torch.optim.Adam + torch.amp.GradScaler, the optimizer pair of the
reference (main_nerf.py:223, nerf/utils.py:1474-1482)."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, N

pytestmark = pytest.mark.gpu


def small_net(seed=0):
    from laenerf_amd.network import NeRFNetwork
    torch.manual_seed(seed)
    net = NeRFNetwork(bound=1, log2_hashmap_size=12).to(DEV)
    net.encoder.embeddings.data.uniform_(-0.1, 0.1)
    return net


def test_fused_adam_matches_torch_adam_and_gradscaler():
    """identical gradients into both optimizers for 9 steps: growth after 3 finite steps, one inf step (skip + backoff)"""
    from laenerf_amd.optim import FusedAdam
    a, b = small_net(), small_net()
    fa = FusedAdam(a, param_groups=a.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, init_scale=1024.0, growth_interval=3)
    tb = torch.optim.Adam(b.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    sc = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=3)
    assert a.encoder.shadow is not None and a.encoder.embeddings.grad is None
    gen = torch.Generator(device=DEV).manual_seed(5)
    pa, pb = [p for g in a.get_params(0) for p in g["params"]], [p for g in b.get_params(0) for p in g["params"]]
    for it in range(9):
        sc.scale(torch.zeros((), device=DEV))                       # lazily creates the scale tensor
        scale = sc.get_scale()
        assert fa.get_scale() == scale
        for x, y in zip(pa, pb):
            g = torch.randn(x.shape, device=DEV, generator=gen) * 1e-3
            g[torch.rand(x.shape, device=DEV, generator=gen) < 0.3] = 0        # untouched table entries
            owner = {id(a.encoder.embeddings): a.encoder, id(a.sigma_net.weights): a.sigma_net, id(a.color_net.weights): a.color_net}[id(x)]
            gh = (g * scale).half()                                              # what the operators' backward leaves there
            if it == 4 and x is a.encoder.embeddings:
                gh.view(-1)[17] = float("inf")
            if it % 2 == 0 or x is a.encoder.embeddings:
                owner.shadow.grad_half.copy_(gh.view_as(owner.shadow.grad_half))
            else:
                x.grad = gh.float()                                              # gradient that arrived through plain autograd
            y.grad = gh.float()
        fa.step()
        sc.step(tb); sc.update()
        for x, y in zip(pa, pb):
            # one step moves a weight by ~lr = 1e-2; formulas agree to a few ulp of that update
            assert torch.allclose(x, y, rtol=2e-6, atol=3e-8), (it, float((x - y).abs().max()))
        assert torch.equal(a.encoder.shadow.half, a.encoder.embeddings.detach().half())
        assert all(float(m.shadow.grad_half.abs().sum()) == 0 for m in (a.encoder, a.sigma_net, a.color_net))
        assert all(x.grad is None for x in pa)
    assert fa.steps_taken == 8 and fa.steps_skipped == 1
    assert fa.get_scale() == sc.get_scale()
    for (p, m, v, _, _), y in zip(fa.items, pb):
        st = tb.state[y]
        assert torch.allclose(m, st["exp_avg"], rtol=1e-5, atol=1e-9) and torch.allclose(v, st["exp_avg_sq"], rtol=1e-5, atol=1e-13)
    # torch-layout checkpoint round trip
    sd = fa.state_dict()
    c = small_net()
    fc = FusedAdam(c, param_groups=c.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    fc.load_state_dict(sd)
    assert fc.steps_taken == 8 and torch.equal(fc.items[0][1], fa.items[0][1]) and fc.get_scale() == fa.get_scale()


def test_shadow_follows_external_writes_and_lr_schedule():
    from laenerf_amd.optim import FusedAdam
    net = small_net(1)
    fa = FusedAdam(net, lr=1e-2, grad_scaler=False)
    x = torch.rand(256, 3, device=DEV) * 2 - 1
    with torch.autocast("cuda", dtype=torch.float16):
        y0 = net.encoder(x)
    with torch.no_grad():
        net.encoder.embeddings.mul_(2.0)                               # in-place write (load_state_dict, init): bumps _version
    with torch.autocast("cuda", dtype=torch.float16):
        y1 = net.encoder(x)
    assert torch.allclose(y1.float(), 2 * y0.float(), rtol=2e-3, atol=1e-4)
    net.encoder.embeddings.data.mul_(0.5)                              # writes through .data are invisible to autograd's
    fa.sync_shadows()                                                  # version counter: the documented explicit refresh
    with torch.autocast("cuda", dtype=torch.float16):
        y2 = net.encoder(x)
    assert torch.allclose(y2.float(), y0.float(), rtol=2e-3, atol=1e-4)
    fa.set_lr(0.0)
    before = net.encoder.embeddings.detach().clone()
    with torch.autocast("cuda", dtype=torch.float16):
        net.encoder(x).float().sum().backward()
    assert net.encoder.embeddings.grad is None and float(net.encoder.shadow.grad_half.abs().sum()) > 0
    fa.step()
    assert torch.equal(before, net.encoder.embeddings.detach()) and fa.steps_taken == 1       # lr 0: state moves, weights do not
    assert float(fa.items[0][1].abs().sum()) > 0


def test_training_with_fused_adam_tracks_torch_path():
    """same model, rays and targets: the fused optimizer path and the torch path produce the same loss curve"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.optim import FusedAdam
    from laenerf_amd.renderer import NeRFRenderer
    bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(DEV)
    o, d = S.lego_like_rays(1024, seed=3)
    o, d = torch.from_numpy(o).to(DEV), torch.from_numpy(d).to(DEV)
    gt = torch.rand(1024, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    curves = []
    for fused in (True, False):
        net = small_net(2)
        r = NeRFRenderer(net, bound=1).to(DEV)
        r.density_bitfield = bits
        net.train()
        if fused:
            opt = sc = FusedAdam(net, param_groups=net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        else:
            opt = torch.optim.Adam(net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
            sc = torch.amp.GradScaler("cuda")
        losses = []
        for it in range(12):
            if not fused:
                opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                res = r.render_train(o, d, bg_color=1, perturb=False, max_steps=256)
                loss = torch.nn.functional.mse_loss(res["image"], gt)
            sc.scale(loss).backward()
            if fused:
                opt.step()
            else:
                sc.step(opt); sc.update()
            losses.append(float(loss.detach()))
        curves.append(losses)
    a, b = np.array(curves[0]), np.array(curves[1])
    assert a[-1] < a[0] * 0.95                                         # it trains (random targets: slowly)
    assert np.allclose(a, b, rtol=2e-2), (a, b)


def test_mse_loss_scaled_matches_torch():
    from laenerf_amd.losses import mse_loss_scaled
    from laenerf_amd.optim import FusedAdam
    gen = torch.Generator(device=DEV).manual_seed(3)
    pred = torch.rand(4096, 3, device=DEV, generator=gen, requires_grad=True)
    gt = torch.rand(4096, 3, device=DEV, generator=gen)
    fa = FusedAdam(small_net(), lr=1e-2, init_scale=512.0)
    loss = mse_loss_scaled(pred, gt, fa)
    loss.backward()
    p2 = pred.detach().clone().requires_grad_()
    ref = torch.nn.functional.mse_loss(p2, gt)
    (ref * 512.0).backward()
    assert torch.allclose(loss.unscaled, ref, rtol=1e-6) and torch.allclose(loss, ref * 512.0, rtol=1e-6)
    assert torch.allclose(pred.grad, p2.grad, rtol=1e-6, atol=1e-12)
    # the reference's form: criterion(reduction='none').mean(-1).mean()
    assert torch.allclose(loss.unscaled, torch.nn.MSELoss(reduction="none")(p2, gt).mean(-1).mean(), rtol=1e-6)
    plain = mse_loss_scaled(pred.detach(), gt)                        # no scaler
    assert torch.allclose(plain, ref, rtol=1e-6)
