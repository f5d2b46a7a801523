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
                owner.shadow.unreported = True       # written behind the operators' back: nobody reported a non-finite value,
                                                     # so the optimizer has to scan this accumulator (TableShadow contract)
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


def test_table_gradient_reports_nonfinite_values_itself():
    """the hash-grid backward and the fused head backward OR the optimizer's found_inf word when they store a non-finite
    gradient, so step() scans nothing (no 24 MB read per step, no check launch) -- and still skips the step, backs the scale off and zeroes everything
    exactly like GradScaler when a gradient overflows; a gradient folded in from plain autograd puts the table back into the scan"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.optim import FusedAdam
    from laenerf_amd.renderer import NeRFRenderer
    bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(DEV)
    o, d = S.lego_like_rays(512, seed=3)
    o, d = torch.from_numpy(o).to(DEV), torch.from_numpy(d).to(DEV)
    gt = torch.rand(512, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    net = small_net(2)
    r = NeRFRenderer(net, bound=1).to(DEV)
    r.density_bitfield = bits
    net.train()
    opt = FusedAdam(net, param_groups=net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, init_scale=1024.0)
    sh = net.encoder.shadow
    assert sh.nonfinite_flag == opt.dev_state.data_ptr() + 8
    before = None
    for it in range(4):
        with torch.autocast("cuda", dtype=torch.float16):
            res = r.render_train(o, d, bg_color=1, perturb=False, max_steps=256)
            loss = torch.nn.functional.mse_loss(res["image"], gt)
        if it == 2:
            loss = loss * float("inf")                     # every gradient of this step is inf / nan
            before = [p.detach().clone() for p in net.parameters()]
        opt.scale(loss).backward()
        assert not sh.unreported                           # only the reporting backward wrote the accumulator
        scanned = opt._check_tables(opt._tables())
        assert scanned["n"] == 0                           # neither the table nor the two weight vectors are scanned
        if it == 2:
            assert int(opt.dev_state[2].item()) == 1       # ... because its backward has reported already
            assert not torch.isfinite(sh.grad_half.float()).all()
        opt.step()
        assert float(sh.grad_half.abs().sum()) == 0
        if it == 2:
            assert all(torch.equal(a_, b_.detach()) for a_, b_ in zip(before, net.parameters()))     # skipped: nothing moved
    assert opt.steps_taken == 3 and opt.steps_skipped == 1 and opt.get_scale() == 512.0
    # a gradient that reaches the table through plain autograd is folded in by the optimizer: scanned again, inf still caught
    net.encoder.embeddings.grad = torch.zeros_like(net.encoder.embeddings)
    net.encoder.embeddings.grad.view(-1)[5] = float("inf")
    assert opt._check_tables(opt._tables())["n"] == 1 and sh.unreported
    opt.step()
    assert opt.steps_skipped == 2 and not sh.unreported
    # an overflowed backward that is DISCARDED (zero_grad without a step) must not make the next, clean step be skipped:
    # zero_grad clears found_inf along with the accumulators
    with torch.autocast("cuda", dtype=torch.float16):
        res = r.render_train(o, d, bg_color=1, perturb=False, max_steps=256)
        loss = torch.nn.functional.mse_loss(res["image"], gt) * float("inf")
    opt.scale(loss).backward()
    assert int(opt.dev_state[2].item()) == 1
    opt.zero_grad()
    assert int(opt.dev_state[2].item()) == 0 and float(sh.grad_half.float().abs().nan_to_num(1.0).sum()) == 0
    taken, skipped, scale = opt.steps_taken, opt.steps_skipped, opt.get_scale()
    with torch.autocast("cuda", dtype=torch.float16):
        res = r.render_train(o, d, bg_color=1, perturb=False, max_steps=256)
        loss = torch.nn.functional.mse_loss(res["image"], gt)
    opt.scale(loss).backward()
    opt.step()
    assert opt.steps_taken == taken + 1 and opt.steps_skipped == skipped and opt.get_scale() == scale


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


def test_fused_adam_is_a_torch_optimizer_lambda_lr_and_reference_checkpoint_layout():
    """the reference steps a LambdaLR every iteration (main_nerf.py:239-245, scheduler_update_every_step) and checkpoints
    `optimizer.state_dict()` of torch.optim.Adam over get_params' FOUR groups (network_ff.py:142-154): both must work"""
    from laenerf_amd.optim import FusedAdam
    a, b = small_net(3), small_net(3)
    fa = FusedAdam(a, param_groups=a.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, init_scale=1024.0, growth_interval=1000)
    tb = torch.optim.Adam(b.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    sc = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=1000)
    assert isinstance(fa, torch.optim.Optimizer) and len(fa.param_groups) == 4 and fa.param_groups[2]["params"] == []
    lam = lambda it: 0.1 ** min(it / 5, 1)
    sa, sb = torch.optim.lr_scheduler.LambdaLR(fa, lam), torch.optim.lr_scheduler.LambdaLR(tb, lam)
    gen = torch.Generator(device=DEV).manual_seed(9)
    pa, pb = [p for g in a.get_params(0) for p in g["params"]], [p for g in b.get_params(0) for p in g["params"]]
    for it in range(7):
        sc.scale(torch.zeros((), device=DEV))                        # lazily creates the scale tensor
        for x, y in zip(pa, pb):
            g = (torch.randn(x.shape, device=DEV, generator=gen) * 1e-3 * 1024.0).half().float()    # scaled, fp16-representable
            x.grad = g.clone(); y.grad = g.clone()                   # plain autograd gradients (folded into the accumulators)
        fa.step()
        sc.step(tb); sc.update()
        sa.step(); sb.step()
        assert fa.param_groups[0]["lr"] == pytest.approx(tb.param_groups[0]["lr"])
        for x, y in zip(pa, pb):
            assert torch.allclose(x.detach(), y.detach(), rtol=2e-6, atol=3e-8), (it, float((x.detach() - y.detach()).abs().max()))
    assert float(fa.lrs[0]) == pytest.approx(1e-2 * 0.1, rel=1e-6)    # the device copy followed the schedule
    # checkpoint: torch's own layout, and torch.optim.Adam's checkpoint loads
    sd, ref = fa.state_dict(), tb.state_dict()
    assert [g["params"] for g in sd["param_groups"]] == [g["params"] for g in ref["param_groups"]] == [[0], [1], [], [2]]
    assert set(sd["state"].keys()) == set(ref["state"].keys()) == {0, 1, 2}
    assert set(sd["state"][0].keys()) >= {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 7
    c = small_net(3)
    fc = FusedAdam(c, param_groups=c.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    fc.load_state_dict(ref)                                           # the reference's optimizer checkpoint
    assert fc.steps_taken == 7 and fc.param_groups[3]["lr"] == pytest.approx(tb.param_groups[3]["lr"])
    for (p, m, v, _, _), y in zip(fc.items, pb):
        assert torch.equal(m, tb.state[y]["exp_avg"]) and torch.equal(v, tb.state[y]["exp_avg_sq"])
    assert float(fc.lrs[3]) == pytest.approx(tb.param_groups[3]["lr"], rel=1e-6)


def test_ema_matches_torch_ema_semantics():
    """torch_ema.ExponentialMovingAverage (nerf/utils.py:407-408, 1502-1503, 1205-1215) restated in numpy: warm-up decay
    min(decay, (1 + n) / (10 + n)), shadow -= (1 - decay) * (shadow - param); store / copy_to / restore"""
    from laenerf_amd.optim import EMA, FusedAdam
    net = small_net(4)
    fa = FusedAdam(net, param_groups=net.get_params(1e-2), grad_scaler=False)
    ema = EMA(net.parameters(), decay=0.95)
    params = [p for p in net.parameters()]
    shadow = [N(p).astype(np.float32).copy() for p in params]
    gen = torch.Generator(device=DEV).manual_seed(2)
    for n_upd in range(1, 6):
        with torch.no_grad():
            for p in params:
                p.add_(torch.randn(p.shape, device=DEV, generator=gen) * 1e-2)
        ema.update()
        decay = min(0.95, (1 + n_upd) / (10 + n_upd))
        omd = np.float32(1.0 - decay)
        for s_, p in zip(shadow, params):
            s_ -= (s_ - N(p)) * omd
        for s_, t in zip(shadow, ema.shadow_params):
            assert np.allclose(N(t), s_, rtol=1e-6, atol=1e-8)
    x = torch.rand(256, 3, device=DEV) * 2 - 1
    live = [p.detach().clone() for p in params]
    ema.store(); ema.copy_to()
    for p, t in zip(params, ema.shadow_params):
        assert torch.equal(p.detach(), t)
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        y_ema = net.encoder(x)                                         # the fp16 shadow table followed the swap (version counter)
    assert torch.equal(net.encoder.shadow.half, ema.shadow_params[0].half())
    ema.restore()
    for p, t in zip(params, live):
        assert torch.equal(p.detach(), t)
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        y_live = net.encoder(x)
    assert torch.equal(net.encoder.shadow.half, live[0].half()) and not torch.equal(y_live, y_ema)
    sd = ema.state_dict()
    e2 = EMA(net.parameters(), decay=0.5); e2.load_state_dict(sd)
    assert e2.num_updates == 5 and all(torch.equal(a_, b_) for a_, b_ in zip(e2.shadow_params, ema.shadow_params))


def test_touched_lines_only_update_is_exact():
    """the binned grid backward keeps a bitmap of the 64-byte table lines it ever stored a gradient in; FusedAdam (weight_decay 0)
    neither reads nor writes the other lines -- their gradient and both Adam moments are zero, so torch.optim.Adam would rewrite
    the same values.  Same model, rays and targets with and without the bitmap: identical parameters, moments and shadow after
    several steps; untouched lines exist (the test would be vacuous otherwise) and a checkpoint load turns every bit on"""
    from laenerf_amd import synthetic as S
    from laenerf_amd.optim import FusedAdam
    from laenerf_amd.renderer import NeRFRenderer
    bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(DEV)
    results = []
    for use_bitmap in (True, False):
        net = small_net(3)
        r = NeRFRenderer(net, bound=1).to(DEV)
        r.density_bitfield = bits
        net.train()
        opt = FusedAdam(net, param_groups=net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, init_scale=128.0)
        sh = net.encoder.shadow
        assert sh.touched_lines is not None
        if not use_bitmap:
            sh.touched_lines = None                        # the plain update of every entry
            opt._args = None
        for it in range(5):
            o, d = S.lego_like_rays(512, seed=10 + it)
            o, d = torch.from_numpy(o).to(DEV), torch.from_numpy(d).to(DEV)
            gt = torch.rand(512, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(it))
            with torch.autocast("cuda", dtype=torch.float16):
                res = r.render_train(o, d, bg_color=1, perturb=False, max_steps=256)
                loss = torch.nn.functional.mse_loss(res["image"], gt)
            opt.scale(loss).backward()
            opt.step()
        item = [x for x in opt.items if x[0] is net.encoder.embeddings][0]
        results.append((net.encoder.embeddings.detach().clone(), item[1].clone(), item[2].clone(), sh.half.clone(), sh, opt))
    (p1, m1, v1, h1, sh1, opt1), (p0, m0, v0, h0, _, _) = results
    assert torch.equal(p1, p0) and torch.equal(m1, m0) and torch.equal(v1, v0) and torch.equal(h1, h0)
    n_lines = p1.numel() // 16
    bitsum = sum(bin(int(w) & 0xffffffff).count("1") for w in sh1.touched_lines.cpu().tolist())
    assert 0 < bitsum < n_lines                              # some lines touched, some never
    # every line that holds a non-zero second moment is marked
    nz = (v1.view(-1, 16) != 0).any(dim=1).cpu().numpy()
    words = np.array(sh1.touched_lines.cpu().tolist(), dtype=np.int64) & 0xffffffff
    marked = ((words[np.arange(n_lines) >> 5] >> (np.arange(n_lines) & 31)) & 1).astype(bool)
    assert not (nz & ~marked).any()
    opt1.load_state_dict(opt1.state_dict())
    assert int((sh1.touched_lines != -1).sum()) == 0
