"""-m gpu: MFMA MLP vs the oracle (fp16 storage, fp32 accumulate).  Tolerances: outputs differ from the oracle only
by the MFMA's internal summation order before the fp16 rounding of each stored activation (<= 1 fp16 ulp per layer)."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, N, T, bits_from_half, half_from_bits

pytestmark = pytest.mark.gpu

CASES = [(32, 64, 2), (32, 64, 3), (48, 64, 2), (48, 64, 3), (64, 64, 2),   # (48, 64, *): the fused backward's instantiations with an ODD number of input k-steps (the K = 16 tail hazard, csrc/ffmlp.hip mfma_ksteps)
         (16, 16, 2), (32, 32, 4), (32, 128, 2), (48, 128, 3), (32, 256, 2)]


def close_f16(a, b, rel=4e-3, floor=2e-3):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.abs(a - b).max() <= rel * np.abs(b).max() + floor


@pytest.mark.parametrize("IN,H,NL", CASES)
@pytest.mark.parametrize("B", [128, 1152])
def test_ffmlp_forward_backward(O, IN, H, NL, B):
    from laenerf_amd.backend import ffmlp_backend as F
    rng = np.random.default_rng(IN + H + NL)
    nW = O.ffmlp_num_params(IN, H, NL)
    W = rng.uniform(-np.sqrt(3 / H), np.sqrt(3 / H), nW).astype(np.float32)
    X = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
    Wh, Xh = O.to_f16_bits(W), O.to_f16_bits(X)
    ref_out, ref_fb = O.ffmlp_forward(Xh, Wh, IN, 16, H, NL)
    out = torch.empty(B, 16, device=DEV, dtype=torch.half); fb = torch.empty(NL, B, H, device=DEV, dtype=torch.half)
    # the reference fills forward_buffer (ffmlp.cu:635-671): mode 1 does for every shape; the default mode leaves it untouched for
    # the shapes whose backward recomputes the activations (include/laenerf.h lae_ffmlp_forward) and fills it for the others
    recompute = F.fused_backward_available(IN, H, NL, 0)
    try:
        F.ffmlp_set_mode(1)
        F.ffmlp_forward(half_from_bits(Xh), half_from_bits(Wh), B, IN, 16, H, NL, 0, 6, fb, out)
        assert close_f16(N(fb), O.from_f16_bits(ref_fb)) and close_f16(N(out), O.from_f16_bits(ref_out))
    finally:
        F.ffmlp_set_mode(0)
    out0 = torch.empty_like(out); fb0 = torch.full_like(fb, -7.0)
    F.ffmlp_forward(half_from_bits(Xh), half_from_bits(Wh), B, IN, 16, H, NL, 0, 6, fb0, out0)
    assert close_f16(N(out0), O.from_f16_bits(ref_out))
    assert bool((fb0 == -7.0).all()) if recompute else close_f16(N(fb0), O.from_f16_bits(ref_fb))
    out = out0
    out_i = torch.empty(B, 16, device=DEV, dtype=torch.half)
    F.ffmlp_inference(half_from_bits(Xh), half_from_bits(Wh), B, IN, 16, H, NL, 0, 6, None, out_i)
    assert torch.equal(out_i, out)
    # backward: feed the ORACLE's forward buffer so both sides see identical ReLU masks
    G = (rng.standard_normal((B, 16)) * 0.05).astype(np.float32); Gh = O.to_f16_bits(G)
    ref_gw, ref_gi, ref_bb = O.ffmlp_backward(Gh, Xh, Wh, ref_fb, IN, 16, H, NL, calc_grad_inputs=True)
    bb = torch.empty(NL, B, H, device=DEV, dtype=torch.half); gi = torch.empty(B, IN, device=DEV, dtype=torch.half)
    gw = torch.empty(nW, device=DEV, dtype=torch.half)
    # mode 1: the buffer-faithful path (fills backward_buffer like the reference); mode 0: fused path where available
    # (recomputes the activations, needs neither buffer)
    try:
        for mode in (1, 0, 3):
            F.ffmlp_set_mode(mode)
            fused = mode != 1 and F.fused_backward_available(IN, H, NL, 0)
            gi.zero_(); gw.zero_()
            F.ffmlp_backward(half_from_bits(Gh), half_from_bits(Xh), half_from_bits(Wh), None if fused else half_from_bits(ref_fb),
                             B, IN, 16, H, NL, 0, 6, True, None if fused else bb, gi, gw)
            if not fused:
                assert close_f16(N(bb), O.from_f16_bits(ref_bb), floor=5e-4)
            assert close_f16(N(gi), O.from_f16_bits(ref_gi), floor=5e-4), mode
            assert close_f16(N(gw), O.from_f16_bits(ref_gw), rel=1e-2, floor=2e-3), mode
            # determinism of the slab reduction
            gw2 = torch.empty_like(gw)
            F.ffmlp_backward(half_from_bits(Gh), half_from_bits(Xh), half_from_bits(Wh), None if fused else half_from_bits(ref_fb),
                             B, IN, 16, H, NL, 0, 6, False, None if fused else bb, gi, gw2)
            assert torch.equal(gw, gw2), mode
    finally:
        F.ffmlp_set_mode(0)


@pytest.mark.parametrize("act", [0, 3, 6])
def test_ffmlp_activations(O, act):
    from laenerf_amd.backend import ffmlp_backend as F
    rng = np.random.default_rng(act)
    IN, H, NL, B = 32, 64, 2, 256
    nW = O.ffmlp_num_params(IN, H, NL)
    Wh = O.to_f16_bits(rng.uniform(-0.2, 0.2, nW).astype(np.float32)); Xh = O.to_f16_bits(rng.uniform(-1, 1, (B, IN)).astype(np.float32))
    ref_out, ref_fb = O.ffmlp_forward(Xh, Wh, IN, 16, H, NL, activation=act)
    out = torch.empty(B, 16, device=DEV, dtype=torch.half); fb = torch.empty(NL, B, H, device=DEV, dtype=torch.half)
    try:
        F.ffmlp_set_mode(1 if act == 0 else 0)           # ReLU at this shape: only mode 1 fills forward_buffer (include/laenerf.h)
        F.ffmlp_forward(half_from_bits(Xh), half_from_bits(Wh), B, IN, 16, H, NL, act, 6, fb, out)
    finally:
        F.ffmlp_set_mode(0)
    assert close_f16(N(out), O.from_f16_bits(ref_out)) and close_f16(N(fb), O.from_f16_bits(ref_fb))


def test_ffmlp_module_matches_linear_chain(O):
    """FFMLP module under autocast == the nn.Linear chain of the reference (golden vectors from nerf/network.py)"""
    from conftest import golden
    from test_oracle_golden_cpu import ff_weights_from_linear
    from laenerf_amd.ffmlp import FFMLP
    g = golden("mlp_chain")
    m = FFMLP(32, 16, 64, 2).to(DEV)
    m.weights.data = T(ff_weights_from_linear([g["sigma_w"], g["sigma_w1"], g["sigma_w2"]], 32))
    with torch.autocast("cuda", dtype=torch.float16):
        y = m(T(g["enc"]))
    assert y.shape == (512, 16) and np.abs(N(y) - g["sigma_h"]).max() < 2e-3
    # training mode + backward through the autograd.Function
    x = T(g["enc"]).requires_grad_()
    with torch.autocast("cuda", dtype=torch.float16):
        y = m.train()(x)
    y.float().square().sum().backward()
    lin = [torch.nn.Linear(32, 64, bias=False), torch.nn.Linear(64, 64, bias=False), torch.nn.Linear(64, 16, bias=False)]
    for l, w in zip(lin, (g["sigma_w"], g["sigma_w1"], g["sigma_w2"])):
        l.weight.data = T(w)
    xr = T(g["enc"]).requires_grad_()
    h = torch.relu(lin[0].to(DEV)(xr)); h = torch.relu(lin[1].to(DEV)(h)); yr = lin[2].to(DEV)(h)
    yr.square().sum().backward()
    gw_ref = torch.cat([l.weight.grad.reshape(-1) for l in lin])
    assert np.abs(N(m.weights.grad) - N(gw_ref)).max() < 2e-2 * N(gw_ref).max()
    assert np.abs(N(x.grad) - N(xr.grad)).max() < 2e-2 * np.abs(N(xr.grad)).max() + 1e-3


@pytest.mark.parametrize("M", [128, 4096 + 16])
def test_nerf_head_matches_operator_chain_and_oracle(O, M):
    """fused head (one forward kernel, two backward kernels) == sigma_net -> trunc_exp/geo_feat -> SH(4) -> cat ->
    color_net -> sigmoid built from the separate operators, and == the same chain on the CPU oracle."""
    from laenerf_amd.network import NeRFNetwork
    rng = np.random.default_rng(M)
    net = NeRFNetwork(bound=1, log2_hashmap_size=12).to(DEV)
    net.train()
    enc_np = (rng.standard_normal((M, 32)) * 0.3).astype(np.float16)
    d = rng.standard_normal((M, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    gs = (rng.standard_normal(M) * 0.1).astype(np.float32)
    gr = (rng.standard_normal((M, 3)) * 0.1).astype(np.float32)

    def run(fused):
        enc = T(enc_np).requires_grad_(True)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                from laenerf_amd.ffmlp import nerf_head
                sigma, rgb = nerf_head(enc, T(d), net.sigma_net.weights, net.color_net.weights)
            else:
                from laenerf_amd.activation import trunc_exp
                h = net.sigma_net(enc)
                sigma = trunc_exp(h[..., 0])
                geo = h[..., 1:]
                sh = net.encoder_dir(T(d))
                cin = torch.cat([sh.to(geo.dtype), geo, torch.zeros_like(geo[..., :1])], dim=-1)
                rgb = torch.sigmoid(net.color_net(cin))
        (sigma.float() * T(gs)).sum().add((rgb.float() * T(gr)).sum()).backward()
        return (N(sigma.float()), N(rgb.float()), N(enc.grad.float()), N(net.sigma_net.weights.grad), N(net.color_net.weights.grad))

    s1, c1, ge1, gws1, gwc1 = run(True)
    s0, c0, ge0, gws0, gwc0 = run(False)
    assert np.allclose(s1, s0, rtol=2e-3, atol=1e-6)                 # h0 may differ by one fp16 ulp (K=32 vs K=16 MFMA summation order)
    assert np.abs(c1 - c0).max() <= 1.5e-3                            # both round the sigmoid to fp16 (a few ulp apart)
    assert close_f16(ge1, ge0, floor=5e-4) and close_f16(gws1, gws0, rel=1e-2) and close_f16(gwc1, gwc0, rel=1e-2)
    # CPU oracle chain (forward)
    ws_h = O.to_f16_bits(N(net.sigma_net.weights)); wc_h = O.to_f16_bits(N(net.color_net.weights))
    h, _ = O.ffmlp_forward(enc_np.view(np.uint16), ws_h, 32, 16, 64, 2)
    hf = O.from_f16_bits(h)
    sh, _ = O.sh_encode_forward(d, 4)
    cin = O.to_f16_bits(np.concatenate([sh, hf[:, 1:], np.zeros((M, 1), np.float32)], 1))
    oc, _ = O.ffmlp_forward(cin, wc_h, 32, 16, 64, 3)
    assert np.allclose(s1, np.exp(hf[:, 0]), rtol=2e-2, atol=1e-6)    # 1 fp16 ulp of h0 -> ~1e-3 relative in exp
    assert np.abs(c1 - 1 / (1 + np.exp(-O.from_f16_bits(oc)[:, :3]))).max() < 3e-3
    # determinism
    s2, c2, ge2, gws2, gwc2 = run(True)
    assert np.array_equal(s1, s2) and np.array_equal(c1, c2) and np.array_equal(ge1, ge2) and np.array_equal(gwc1, gwc2)


@pytest.mark.parametrize("M", [16, 48, 4096 + 16, 64 * 1031])
@pytest.mark.parametrize("level_major", [False, True])
def test_nerf_head_kernel_variants_agree(M, level_major):
    """the fused head on row counts with an odd number of 16-row tiles, fewer rows than one 64-row group, and more groups than
    waves.  Forward: a row's outputs do not depend on how many other rows share the launch (the first rows evaluated alone give
    the same bits) -- the operator-chain / fixture comparison is test_nerf_head_matches_operator_chain.  Backward: the two
    kernels that ship (wave-private dW with MFMA transposes = default / workgroup-cooperative dW, round 2) give the same input
    gradients bit for bit and weight gradients equal up to fp32 summation order."""
    from laenerf_amd.backend import ffmlp_backend as F
    g = torch.Generator(device=DEV).manual_seed(M)
    enc = ((torch.randn(16, M, 2, device=DEV, generator=g) if level_major else torch.randn(M, 32, device=DEV, generator=g)) * 0.5).half()
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=DEV, generator=g), dim=-1).contiguous()
    ws = ((torch.rand(64 * (32 + 64 + 16), device=DEV, generator=g) * 2 - 1) * 0.2165).half()
    wc = ((torch.rand(64 * (32 + 128 + 16), device=DEV, generator=g) * 2 - 1) * 0.2165).half()
    gs = torch.randn(M, device=DEV, generator=g) * 1e-2
    gr = torch.randn(M, 3, device=DEV, generator=g) * 1e-2

    def fwd(enc_, dirs_, m):
        h = torch.full((m, 16), float("nan"), dtype=torch.half, device=DEV)
        sig = torch.full((m,), float("nan"), device=DEV); rgb = torch.full((m, 3), float("nan"), device=DEV)
        F.nerf_head_forward(enc_, dirs_, ws, wc, m, 1.0, h, sig, rgb, level_major=level_major)
        return h, sig, rgb
    h, sig, rgb = fwd(enc, dirs, M)
    assert torch.isfinite(h.float()).all() and torch.isfinite(sig).all() and torch.isfinite(rgb).all()
    m = min(M, 32)
    sub = enc[:, :m].contiguous() if level_major else enc[:m].contiguous()
    h2, sig2, rgb2 = fwd(sub, dirs[:m].contiguous(), m)
    assert torch.equal(h2, h[:m]) and torch.equal(sig2, sig[:m]) and torch.equal(rgb2, rgb[:m])
    try:
        grads = []
        for mode in (3, 0):
            F.ffmlp_set_mode(mode)
            gh = torch.full((M, 16), float("nan"), dtype=torch.half, device=DEV)
            genc = torch.full(enc.shape, float("nan"), dtype=torch.half, device=DEV)
            gws, gwc = torch.zeros_like(ws), torch.zeros_like(wc)
            F.nerf_head_backward(gs, gr, enc, dirs, h, rgb, ws, wc, M, 1.0, gh, genc, gws, gwc, accumulate=False, level_major=level_major)
            grads.append((gh, genc, gws, gwc))
        for gh, genc, gws, gwc in grads[1:]:
            assert torch.equal(gh, grads[0][0]) and torch.equal(genc, grads[0][1])
            for a, b in ((gws, grads[0][2]), (gwc, grads[0][3])):
                assert float((a.float() - b.float()).abs().max()) <= 2e-3 * float(b.float().abs().max()) + 1e-7
    finally:
        F.ffmlp_set_mode(0)


def test_nerf_head_rejects_bad_shapes():
    from laenerf_amd.ffmlp import nerf_head
    w1 = torch.zeros(64 * 112, device=DEV); w2 = torch.zeros(64 * 176, device=DEV)
    with pytest.raises(RuntimeError):
        nerf_head(torch.zeros(24, 32, device=DEV, dtype=torch.half), torch.zeros(24, 3, device=DEV), w1, w2)
    with pytest.raises(RuntimeError):
        nerf_head(torch.zeros(32, 16, device=DEV, dtype=torch.half), torch.zeros(32, 3, device=DEV), w1, w2)


def test_nerf_head_weight_gradients_accumulate_into_shadow_buffers():
    """with a FusedAdam attached the head adds dW into the optimizer's persistent fp16 buffers: two backward passes
    before one step() must leave twice the gradient of one"""
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = NeRFNetwork(bound=1, log2_hashmap_size=12).to(DEV)
    net.train()
    FusedAdam(net, lr=1e-2, grad_scaler=False)
    x = torch.rand(2048, 3, device=DEV) * 1.6 - 0.8
    d = torch.nn.functional.normalize(torch.randn(2048, 3, device=DEV), dim=-1)

    def backward_once():
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb = net(x, d)
        (sigma.mean() + rgb.mean()).backward()

    backward_once()
    one = [m.shadow.grad_half.float().clone() for m in (net.sigma_net, net.color_net, net.encoder)]
    assert all(float(g.abs().sum()) > 0 for g in one) and net.sigma_net.weights.grad is None
    backward_once()
    for g1, m in zip(one, (net.sigma_net, net.color_net, net.encoder)):
        assert torch.allclose(m.shadow.grad_half.float(), 2 * g1, rtol=2e-3, atol=1e-6)


def test_nerf_field_equals_encoder_then_head():
    """encoder + head as one op (level-major features between the kernels) == GridEncoder followed by nerf_head:
    same kernels, same arithmetic -> identical outputs and identical gradients"""
    from laenerf_amd.network import NeRFNetwork
    torch.manual_seed(3)
    net = NeRFNetwork(bound=2, log2_hashmap_size=14).to(DEV)
    net.encoder.embeddings.data.uniform_(-0.3, 0.3)
    net.train()
    M = 4096 + 128
    x = (torch.rand(M, 3, device=DEV) * 2 - 1) * 2.0
    x[:4] = torch.tensor([[2.0, 2.0, 2.0], [-2.0, -2.0, -2.0], [2.1, 0, 0], [0, 0, 0]], device=DEV)   # corners, out of range
    d = torch.nn.functional.normalize(torch.randn(M, 3, device=DEV), dim=-1)
    gs, gr = torch.randn(M, device=DEV) * 0.1, torch.randn(M, 3, device=DEV) * 0.1
    res = []
    for fused in (True, False):
        net.fused_field = fused
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb = net(x, d)
        ((sigma * gs).sum() + (rgb * gr).sum()).backward()
        res.append((sigma.detach().clone(), rgb.detach().clone(), net.encoder.embeddings.grad.clone(),
                    net.sigma_net.weights.grad.clone(), net.color_net.weights.grad.clone()))
    for k, (a, b) in enumerate(zip(*res)):
        if k == 2:      # table gradient: levels smaller than 16 partitions are split into sub-buckets whose partial sums meet
            assert torch.allclose(a, b, rtol=1e-2, atol=2e-3 * float(b.abs().max()))    # in fp16 atomics (order varies)
        else:
            assert torch.equal(a, b), k
    assert float(res[0][2].abs().sum()) > 0
