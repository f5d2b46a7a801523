"""-m gpu: LAENeRF's palette network (laenerf_amd/editing/style_encoder.py, csrc/palette.hip) against the fp32 torch
formulation of editing/style_encoder.py:135-158 (the reference's own MLPs are tinycudann, un-vendored: parity unpinned by
execution, anchored on the nn.Linear-equivalent chain of the same shapes)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from gpu_util import DEV, N

pytestmark = pytest.mark.gpu

# palette network against torch's half-precision linear chain on the same parameters (test_laenerf_forward_train_and_gradients):
# gradients relative to the largest reference entry (table: relative L2), outputs absolute.  Rounds 1-3 compared with an fp32 chain
# at 1e-2 / 8 % / 10 %.
STYLE_TOL = {"pred": 1e-3, "w_hat": 1e-4, "o_hat": 5e-4, "loss": 1e-5, "g_wn": 1e-3, "g_on": 1e-3, "g_pal": 2e-3, "g_table": 1e-3}
# observed (n = 4096 / 1000): pred 5.0e-4 / 3.5e-4 (one fp16 ulp at 0.5-1), w_hat 3.5e-5, o_hat 2.4e-4, loss 2.6e-6, g_wn 2.5e-4 / 6e-5,
# g_on 1.1e-4 / 4.4e-4, g_pal 7.6e-4 / 2.9e-4, g_table 3.4e-4 / 4.6e-4


def torch_recompose(w_logits, o_raw, palette, active):
    """style_encoder.py:148-158 in fp32"""
    w_hat = torch.softmax(w_logits[:, :palette.shape[0]][:, active].float(), -1)
    o_hat = torch.tanh(o_raw[:, :3].float())
    pre = w_hat @ palette[active].half().float() + o_hat
    return torch.clamp(pre, 0, 1), w_hat, o_hat


@pytest.mark.parametrize("P,mask,M", [(8, 0xFF, 1000), (8, 0b10110101, 777), (5, 0b11111, 64), (16, 0xFFFF, 300), (3, 0b100, 130)])
def test_palette_recompose_forward_backward(P, mask, M):
    from laenerf_amd.editing import palette_recompose
    torch.manual_seed(P * 1000 + M)
    wl = (torch.randn(M, 16, device=DEV) * 2).half().requires_grad_(True)
    ol = (torch.randn(M, 16, device=DEV) * 1.5).half().requires_grad_(True)
    pal = torch.rand(P, 3, device=DEV).requires_grad_(True)
    active = torch.tensor([(mask >> k) & 1 == 1 for k in range(P)], device=DEV)
    pred, w_hat, o_hat = palette_recompose(wl, ol, pal, mask)
    wl2, ol2, pal2 = wl.detach().clone().requires_grad_(True), ol.detach().clone().requires_grad_(True), pal.detach().clone().requires_grad_(True)
    rp, rw, ro = torch_recompose(wl2, ol2, pal2, active)
    assert w_hat.shape == rw.shape and w_hat.dtype == torch.float32 and pred.dtype == torch.half
    assert np.abs(N(w_hat) - N(rw)).max() < 2e-6
    assert np.abs(N(o_hat) - N(ro)).max() < 1e-3                       # fp16 rounding of tanh
    assert np.abs(N(pred) - N(rp)).max() < 2e-3
    assert np.allclose(N(w_hat).sum(-1), 1, atol=1e-5)
    # backward with all three outputs in the loss, moderate gradients (fp16 storage of dL/dlogits)
    gp, gw, go = torch.randn_like(rp), torch.randn_like(rw), torch.randn_like(ro)
    (pred.float() * gp).sum().add((w_hat * gw).sum()).add((o_hat.float() * go).sum()).backward()
    ((rp * gp).sum() + (rw * gw).sum() + (ro * go).sum()).backward()
    # the clamp mask is evaluated on fp16-rounded values here and on fp32 values in the torch chain: compare rows
    # whose pre-clamp value is away from 0 and 1
    pre = (rw @ pal2[active].half().float() + ro).detach()
    safe = ((pre - 0).abs() > 5e-3).all(-1) & ((pre - 1).abs() > 5e-3).all(-1)
    assert safe.float().mean() > 0.9
    s = N(safe)
    assert np.abs(N(wl.grad)[s] - N(wl2.grad)[s]).max() < 2e-2
    assert np.abs(N(ol.grad)[s] - N(ol2.grad)[s]).max() < 2e-2
    assert (N(wl.grad)[:, P:] == 0).all() and (N(ol.grad)[:, 3:] == 0).all()
    assert (N(wl.grad)[:, :P][:, ~N(active)] == 0).all()
    if bool(safe.all()):
        assert np.abs(N(pal.grad) - N(pal2.grad)).max() < 2e-2 * max(1.0, np.abs(N(pal2.grad)).max())
    assert (N(pal.grad)[~N(active)] == 0).all()
    # deterministic
    wl.grad = None; ol.grad = None; pal.grad = None
    pred, w_hat, o_hat = palette_recompose(wl, ol, pal, mask)
    (pred.float() * gp).sum().add((w_hat * gw).sum()).add((o_hat.float() * go).sum()).backward()
    g1 = pal.grad.clone(); pal.grad = None; wl.grad = None; ol.grad = None
    pred, w_hat, o_hat = palette_recompose(wl, ol, pal, mask)
    (pred.float() * gp).sum().add((w_hat * gw).sum()).add((o_hat.float() * go).sum()).backward()
    assert torch.equal(g1, pal.grad)


def chain(x, W, dims):
    """bias-free ReLU MLP from the FFMLP flat layout W0[h,in] | Wk[h,h] | Wout[16,h]"""
    off = 0
    h = x
    for i, (o, inn) in enumerate(dims):
        w = W[off:off + o * inn].view(o, inn); off += o * inn
        h = h @ w.t()
        if i != len(dims) - 1:
            h = torch.relu(h)
    return h


def make_model(P=8, dir_encoding="sphere_harmonics"):
    from laenerf_amd.editing import LAENeRF
    params = SimpleNamespace(bound=1, num_palette_bases=P, style_weight=0, weight_loss_uniform=1e-3, weight_loss_non_uniform=1e-3,
                             offset_loss=1e-2, palette_loss_valid=1.0, palette_loss_distinct=1e-2)
    torch.manual_seed(0)
    m = LAENeRF(params, dir_encoding=dir_encoding).to(DEV)
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)
    m.weight_net.weights.data.uniform_(-0.3, 0.3)
    m.offset_net.weights.data.uniform_(-0.3, 0.3)
    return m, params


def reference_forward(m, x, d, half=False):
    """restatement of style_encoder.py:135-158 on the same parameters.  half=False: fp32 `nn.Linear`-shaped chain (the shape
    anchor of rounds 1-3).  half=True: the SAME PRECISION PATH as the kernels -- what torch's own nn.Linear chain does under fp16
    autocast (the reference's default nets, nerf/network.py:95-124: half operands, fp32 accumulate, every layer's output and every
    layer's gradient rounded to half), so that outputs and gradients can be pinned at rounding level instead of at 8-10 %
    (VERDICT r3 weak 1 iii; the e2e pins went the same way in round 3)."""
    feat = m.encoder(x, bound=m.bound)
    feat = feat.half() if half else feat.float()
    cast = (lambda w: w.half()) if half else (lambda w: w.float())
    wl = chain(feat, cast(m.weight_net.weights), [(64, 32), (64, 64), (16, 64)])
    cols = [feat]
    if m.dir_encoding is not None:
        cols.append(m.dir_encoding(d).to(feat.dtype))
    cols.append(feat.new_zeros(feat.shape[0], m.offset_in_dim - sum(c.shape[1] for c in cols)))
    ol = chain(torch.cat(cols, -1), cast(m.offset_net.weights), [(64, m.offset_in_dim), (64, 64), (16, 64)])
    return torch_recompose(wl, ol, m.color_palette.float(), m.active_palets)


@pytest.mark.parametrize("n", [4096, 1000])
def test_laenerf_forward_train_and_gradients(n):
    m, params = make_model()
    m.train()
    torch.manual_seed(1)
    x = (torch.rand(n, 3, device=DEV) * 2 - 1) * 0.3
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV), dim=-1)
    target = torch.rand(n, 3, device=DEV)
    assert m.offset_in_dim == 48 and m.weight_net.num_layers == 2

    def loss_of(pred, w, o):
        return torch.nn.functional.mse_loss(pred.float(), target) + m.weights_loss(w.float(), params) + m.offset_loss(o.float(), params) \
            + m.palet_loss(params)
    with torch.autocast("cuda", dtype=torch.float16):
        pred, w_hat, o_hat = m.forward_train(x, d)
        loss = loss_of(pred, w_hat, o_hat)
    assert pred.shape == (n, 3) and w_hat.shape == (n, 8) and o_hat.shape == (n, 3)
    (loss * 128.0).backward()
    got = {k: (v.grad / 128.0).clone() for k, v in (("table", m.encoder.embeddings), ("wn", m.weight_net.weights),
                                                     ("on", m.offset_net.weights), ("pal", m.color_palette))}
    m.zero_grad()
    # (a) shape anchor: the fp32 chain (loose: it differs from the kernels by the fp16 rounding of every activation)
    rp, rw, ro = reference_forward(m, x, d)
    assert np.abs(N(pred) - N(rp)).max() < 1e-2 and np.abs(N(w_hat) - N(rw)).max() < 1e-2 and np.abs(N(o_hat) - N(ro)).max() < 1e-2
    # (b) the same precision path (torch half linears: fp32 accumulate, half outputs and half gradients per layer, same loss scale)
    with torch.autocast("cuda", dtype=torch.float16):
        hp, hw, ho = reference_forward(m, x, d, half=True)
        hloss = loss_of(hp, hw, ho)
    dev = {"pred": float(np.abs(N(pred) - N(hp)).max()), "w_hat": float(np.abs(N(w_hat) - N(hw)).max()), "o_hat": float(np.abs(N(o_hat) - N(ho)).max()),
           "loss": abs(loss.item() - hloss.item()) / abs(hloss.item())}
    (hloss * 128.0).backward()
    for k, p in (("wn", m.weight_net.weights), ("on", m.offset_net.weights), ("pal", m.color_palette)):
        ref = N(p.grad) / 128.0
        dev["g_" + k] = float(np.abs(N(got[k]) - ref).max() / np.abs(ref).max())
    gt, rt = N(got["table"]), N(m.encoder.embeddings.grad) / 128.0
    dev["g_table"] = float(np.linalg.norm(gt - rt) / np.linalg.norm(rt))
    print("style net vs torch half chain:", {k: float("%.3g" % v) for k, v in dev.items()})
    for k, v in dev.items():
        assert v < STYLE_TOL[k], dev


def test_laenerf_inference_active_palettes_and_no_dirs():
    m, params = make_model(P=6)
    m.eval()
    x = (torch.rand(500, 3, device=DEV) * 2 - 1) * 0.3
    d = torch.nn.functional.normalize(torch.randn(500, 3, device=DEV), dim=-1)
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        full = m(x, d)
        m.set_active_palets([True, False, True, True, False, True])
        part, w, o = m.forward_train(x, d)
        assert w.shape == (500, 4) and m.get_color_palette().shape == (4, 3)
        rp, rw, ro = reference_forward(m, x, d)
        assert np.abs(N(part) - N(rp)).max() < 1e-2 and np.abs(N(w) - N(rw)).max() < 1e-2
        gw = m.get_weights(x)
        assert np.abs(N(gw) - N(w)).max() < 2e-3
        assert np.abs(N(m.get_offsets(x, d)).astype(np.float32) - np.arctanh(np.clip(N(ro), -0.999, 0.999))).max() < 5e-2
        m.distill_color_palettes([x, x * 0.5], n=4, thresh=1e-6)      # switched-off bases have mean weight 0
        assert int(m.active_palets.sum()) == 4 and m._active_mask == 0b101101
        with pytest.raises(ValueError):
            m.set_active_palets([False] * 6)
    assert full.shape == (500, 3)
    m2, _ = make_model(dir_encoding=None)
    assert m2.offset_in_dim == 32
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        p2 = m2(x)
        rp2 = reference_forward(m2, x, None)[0]
    assert np.abs(N(p2) - N(rp2)).max() < 1e-2


@pytest.mark.parametrize("palet", [False, True])
@pytest.mark.parametrize("n,mask", [(4096, 0xFF), (1008, 0b10110101)])
def test_fused_point_losses_equal_the_torch_formulation(n, mask, palet):
    """forward_train_loss == forward_train followed by MSE + weights_loss + offset_loss [+ palet_loss, the palette-only regulariser
    of style_encoder.py:195-202, with_palet_loss=True] in torch (value and gradients).  The palette starts with one base outside
    [0, 1) so that floor(p) * p and its gradient are not identically zero."""
    m, params = make_model()
    m.set_active_palets([(mask >> k) & 1 == 1 for k in range(8)])
    m.train()
    torch.manual_seed(2)
    x = (torch.rand(n, 3, device=DEV) * 2 - 1) * 0.3
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV), dim=-1)
    target = torch.rand(n, 3, device=DEV)
    scale = torch.tensor([256.0], device=DEV)
    with torch.no_grad():
        m.color_palette[1] = torch.tensor([1.3, -0.4, 0.7], device=DEV)
    res = []
    for fused in (False, True):
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                loss, pred, w, o = m.forward_train_loss(x, d, target, params, scale, with_palet_loss=palet)
                if palet:
                    assert loss.terms[8].item() == pytest.approx(m.palet_loss(params).item(), rel=1e-5)
            else:
                pred, w, o = m.forward_train(x, d)
                # nerf/utils.py:990-995: every added term goes through .half() (identity in the backward)
                loss = torch.nn.functional.mse_loss(pred.float(), target) + m.weights_loss(w, params).half() + m.offset_loss(o.float(), params).half()
                if palet:
                    loss = loss + m.palet_loss(params).half()
                loss = loss * scale
        (loss * 0.5).backward()
        res.append((loss.detach().float().clone(), pred.detach().clone(), [p.grad.clone() for p in (m.color_palette, m.weight_net.weights,
                                                                                                     m.offset_net.weights, m.encoder.embeddings)]))
    (l0, p0, g0), (l1, p1, g1) = res
    assert torch.equal(p0, p1)
    assert l1.item() == pytest.approx(l0.item(), rel=2e-5)          # the fp16 roundings of the added terms are the reference's; was 2e-4
    terms = res[1][0]
    for a, b, name in zip(g0, g1, ("palette", "weight_net", "offset_net", "table")):
        a, b = N(a), N(b)
        assert np.abs(a - b).max() <= 0.02 * np.abs(a).max() + 1e-6, name      # fp16 rounding of dL/dlogits in both paths


def test_palette_and_loss_argument_errors():
    """error behaviour of the C ABI through the Python stub: invalid arguments raise RuntimeError (no silent fallback)"""
    from laenerf_amd.backend import style_backend as B
    from laenerf_amd.editing import palette_recompose
    wl = torch.zeros(32, 16, device=DEV, dtype=torch.half)
    pal = torch.rand(8, 3, device=DEV)
    with pytest.raises(RuntimeError):
        palette_recompose(wl, wl, pal, 0)                                   # no active palette base
    with pytest.raises(RuntimeError):
        palette_recompose(wl, wl, torch.rand(17, 3, device=DEV), 0x1FFFF)  # more bases than the 16-wide MLP output
    with pytest.raises(RuntimeError):
        palette_recompose(wl.cpu(), wl.cpu(), pal.cpu(), 0xFF)              # CPU tensors: no fallback
    pred = torch.zeros(32, 3, device=DEV, dtype=torch.half)
    w = torch.zeros(32, 8, device=DEV)
    fin = torch.zeros(12, device=DEV)
    with pytest.raises(RuntimeError):
        B.style_loss_forward(pred, torch.zeros(32, 3, device=DEV), w, pred, 32, 0, (1, 1, 1), None, fin)     # n_active = 0
    # empty batch: forward is a no-op, backward zeroes the palette gradient
    e = torch.zeros(0, 16, device=DEV, dtype=torch.half, requires_grad=True)
    p2 = pal.clone().requires_grad_(True)
    a, b, c = palette_recompose(e, e, p2, 0xFF)
    assert a.shape == (0, 3) and b.shape == (0, 8)
    (a.float().sum() + b.sum() + c.float().sum()).backward()
    assert torch.equal(p2.grad, torch.zeros_like(p2))


@pytest.mark.parametrize("n,dirs", [(4096, True), (5003, True), (2000, False)])
def test_fused_input_assembly_equals_the_operator_chain(n, dirs):
    """round 5: `_style_features` (level-major encoder output -> lae_style_assemble_forward -> the two MLPs' input rows; backward:
    the two input gradients added and transposed back in one launch) against the operator chain it replaces (GridEncoder rows, SHEncoder,
    cast, zero pad, cat; slice copies, add, transpose) -- and the shadow-aware FFMLP backward (weight gradients ADDED to the optimizer's
    fp16 accumulators, non-finite values reported by the kernel) against the autograd `.grad` + fold path: logits, prediction, the
    table gradient and both MLPs' weight gradients BIT for bit, with a FusedAdam attached (the shipped configuration), and the
    parameters after one optimizer step."""
    from laenerf_amd.optim import FusedAdam
    out = []
    for new_path in (True, False):
        m, _ = make_model(dir_encoding="sphere_harmonics" if dirs else None)
        m.train()
        m.fused_inputs = m.ffmlp_shadows = new_path
        opt = FusedAdam(m, param_groups=m.get_params(1e-3), betas=(0.9, 0.999), eps=1e-8, init_scale=128.0)
        g = torch.Generator(device=DEV).manual_seed(n)
        x = (torch.rand(n, 3, device=DEV, generator=g) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=g), dim=-1) if dirs else None
        with torch.autocast("cuda", dtype=torch.float16):
            assert m._fused_inputs_ok(x, d) is new_path
            w_logits, o_raw, M = m._logits(x, d)
            pred, w_hat, o_hat = m.forward_train(x, d)
            loss = ((pred.float() ** 2).mean() + (w_hat ** 2).mean() + (o_hat.float() ** 2).mean()) * opt._scale_view[0]
        opt.backward(loss)
        grads = [m.encoder.shadow.grad_half.clone(), m.weight_net.shadow.grad_half.clone(), m.offset_net.shadow.grad_half.clone()]
        for k, net in ((1, m.weight_net), (2, m.offset_net)):
            if net.weights.grad is not None:           # the old path leaves the MLP gradients in .grad: fold them like FusedAdam._grad does
                assert not new_path
                grads[k] = grads[k] + net.weights.grad.to(torch.half)
            else:
                assert new_path
        opt.step()
        out.append((w_logits.detach().clone(), o_raw.detach().clone(), pred.detach().clone(), grads,
                    [p_.detach().clone() for p_ in (m.encoder.embeddings, m.weight_net.weights, m.offset_net.weights, m.color_palette)]))
    a, b = out
    assert a[0].shape == b[0].shape == ((n + 15) // 16 * 16, 16) and M == n
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for ga, gb, name in zip(a[3], b[3], ("table", "weight_net", "offset_net")):
        assert torch.equal(ga, gb), (name, float((ga.float() - gb.float()).abs().max()))
        assert float(ga.float().abs().sum()) > 0
    for pa, pb in zip(a[4], b[4]):
        assert torch.equal(pa, pb)


def test_palette_gradient_added_inside_the_criterion_equals_autograd_accumulation():
    """round 5: with a FusedAdam attached the fused criterion adds the palette gradient into the optimizer's persistent fp32 `.grad`
    inside its reduction launch (LAE_STYLE_ACCUMULATE_PALETTE) instead of handing it to autograd's AccumulateGrad: same values bit
    for bit after one and after two backward passes (fp32 adds in the same order), and the buffer keeps its address."""
    from laenerf_amd.optim import FusedAdam
    got = []
    for direct in (True, False):
        m, params = make_model()
        m.train()
        opt = FusedAdam(m, param_groups=m.get_params(1e-3), betas=(0.9, 0.999), eps=1e-8, init_scale=64.0)
        assert m.color_palette._lae_persistent_grad
        if not direct:
            del m.color_palette._lae_persistent_grad
        addr = m.color_palette.grad.data_ptr()
        g = torch.Generator(device=DEV).manual_seed(3)
        x = (torch.rand(4096, 3, device=DEV, generator=g) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(4096, 3, device=DEV, generator=g), dim=-1)
        target = torch.rand(4096, 3, device=DEV, generator=g)
        snaps = []
        for _ in range(2):
            with torch.autocast("cuda", dtype=torch.float16):
                loss, *_rest = m.forward_train_loss(x, d, target, params, opt, with_palet_loss=True)
            opt.backward(loss)
            assert m.color_palette.grad.data_ptr() == addr
            snaps.append(m.color_palette.grad.clone())
        got.append(snaps)
    for a, b in zip(*got):
        assert torch.equal(a, b) and float(a.abs().sum()) > 0
    assert not torch.equal(got[0][0], got[0][1])


@pytest.mark.parametrize("n", [4096, 5008])
def test_step_with_a_backward_plan_made_ahead_equals_the_step_that_plans_for_itself(n):
    """round 5: LAENeRF.plan_backward(x) (counting half of the hash-grid backward, positions only) made on ANOTHER stream before
    the step, handed to forward_train_loss(plan=...): table gradient, touched-lines bitmap and every parameter after the optimizer
    step bit for bit equal to the step without a plan; a plan without the one-node input path is refused."""
    from laenerf_amd.optim import FusedAdam
    out = []
    for planned in (True, False):
        m, params = make_model()
        m.train()
        opt = FusedAdam(m, param_groups=m.get_params(1e-3), betas=(0.9, 0.999), eps=1e-8, init_scale=128.0)
        g = torch.Generator(device=DEV).manual_seed(n)
        x = (torch.rand(n, 3, device=DEV, generator=g) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=g), dim=-1)
        target = torch.rand(n, 3, device=DEV, generator=g)
        plan = None
        if planned:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                plan = m.plan_backward(x)
            assert plan is not None and plan.marks_touched
            torch.cuda.current_stream().wait_stream(side)
            first_plan = plan
        with torch.autocast("cuda", dtype=torch.float16):
            loss, *_rest = m.forward_train_loss(x, d, target, params, opt, with_palet_loss=True, plan=plan)
        opt.backward(loss)
        grad = m.encoder.shadow.grad_half.clone()
        touched = m.encoder.shadow.touched_lines.clone()
        opt.step()
        out.append([loss.detach().clone(), grad, touched] + [p_.detach().clone() for p_ in m.parameters()])
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert float(out[0][1].float().abs().sum()) > 0
    m.fused_inputs = False
    assert m.plan_backward(x) is None
    with torch.autocast("cuda", dtype=torch.float16), pytest.raises(RuntimeError):
        m.forward_train_loss(x, d, target, params, opt, plan=first_plan)


def test_round5_entry_points_refuse_bad_arguments():
    """error behaviour of the round-5 C entry points through the Python stubs: wrong shapes / dtypes / modes raise (no fallback)"""
    from laenerf_amd.backend import style_backend as B, ffmlp_backend as F, raymarching_backend as R
    M, Mp = 100, 112
    feats = torch.zeros(16, M, 2, device=DEV, dtype=torch.half)
    feat = torch.zeros(Mp, 32, device=DEV, dtype=torch.half)
    off = torch.zeros(Mp, 48, device=DEV, dtype=torch.half)
    dirs = torch.zeros(M, 3, device=DEV)
    with pytest.raises(RuntimeError):
        B.style_assemble_forward(feats, dirs, M, Mp, 5, feat, off, 48)                 # SH degree above 4
    with pytest.raises(RuntimeError):
        B.style_assemble_forward(feats, dirs, M, Mp, 4, feat, off[:, :40].contiguous(), 40)     # 32 + 16 columns do not fit 40
    with pytest.raises(RuntimeError):
        B.style_assemble_forward(feats, dirs, M, M - 4, 3, feat, off, 48)              # padded rows below the rows
    with pytest.raises(RuntimeError):
        B.style_assemble_forward(feats.float(), dirs, M, Mp, 3, feat, off, 48)         # fp32 features
    with pytest.raises(RuntimeError):
        B.style_assemble_backward(None, None, M, 48, feats)                            # no gradient at all
    with pytest.raises(RuntimeError):
        B.style_assemble_backward(feat, off, M, 50, feats)                             # more than 48 columns
    B.style_assemble_forward(feats, dirs, M, Mp, 3, feat, off, 48)                     # the valid call goes through
    B.style_assemble_backward(feat, off, M, 48, feats)
    # accumulate / nonfinite_flag need the fused backward: a width it does not cover is refused, not silently overwritten
    w = torch.zeros(32 * 16 + 16 * 16 + 16 * 16, device=DEV, dtype=torch.half)
    x = torch.zeros(16, 32, device=DEV, dtype=torch.half)
    g = torch.zeros(16, 16, device=DEV, dtype=torch.half)
    if not F.fused_backward_available(32, 16, 2, 0):
        fb = torch.zeros(2, 16, 16, device=DEV, dtype=torch.half)
        with pytest.raises(RuntimeError):
            F.ffmlp_backward(g, x, w, fb, 16, 32, 16, 16, 2, 0, 6, False, torch.zeros_like(fb), torch.zeros(1, device=DEV, dtype=torch.half),
                             torch.zeros_like(w), accumulate=True)
    probe = R.render_frame_probe()
    assert set(probe) == {"in_use", "handshake_us"} and len(probe["handshake_us"]) == 5
    torch.cuda.synchronize()


def test_palette_gradient_reaches_autograd_users_outside_the_fused_backward():
    """the in-kernel add into the palette's persistent `.grad` is taken only inside FusedAdam.backward() and for a
    parameter without hooks; torch.autograd.grad() returns the gradient (and leaves `.grad` alone), a tensor hook fires, and both see
    the values the direct path adds."""
    from laenerf_amd.optim import FusedAdam
    m, params = make_model()
    m.train()
    opt = FusedAdam(m, param_groups=m.get_params(1e-3), betas=(0.9, 0.999), eps=1e-8, init_scale=64.0)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.rand(2048, 3, device=DEV, generator=g) - 0.5) * 1.2
    d = torch.nn.functional.normalize(torch.randn(2048, 3, device=DEV, generator=g), dim=-1)
    tgt = torch.rand(2048, 3, device=DEV, generator=g)

    def loss_of():
        with torch.autocast("cuda", dtype=torch.float16):
            return m.forward_train_loss(x, d, tgt, params, opt, with_palet_loss=True)[0]
    opt.backward(loss_of())                                    # direct path: the criterion adds into .grad itself
    direct = m.color_palette.grad.clone()
    assert float(direct.abs().sum()) > 0
    m.color_palette.grad.zero_()
    before = m.color_palette.grad.clone()
    got, = torch.autograd.grad(loss_of(), m.color_palette)     # not inside FusedAdam.backward: autograd returns it ...
    assert got is not None and torch.equal(got, direct)
    assert torch.equal(m.color_palette.grad, before)           # ... and .grad was not touched
    seen = []
    h = m.color_palette.register_hook(lambda gr: seen.append(gr.clone()))
    try:
        opt.backward(loss_of())                                # a hooked parameter: through AccumulateGrad, the hook fires
    finally:
        h.remove()
    assert len(seen) == 1 and torch.equal(seen[0], direct) and torch.equal(m.color_palette.grad, direct)
