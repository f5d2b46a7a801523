"""The zero-edit drop-in train step (tools/reference_chain.py: the reference's operator sequence and wrapper rules through the
backend modules under their reference names) computes the same step as this repository's fused driver.

Reference: nerf/renderer.py:259-333, nerf/network_ff.py:51-79, gridencoder/grid.py:24-93, ffmlp/ffmlp.py:15-86, nerf/utils.py:1472-1478.
Tolerances: images 1e-4 (north_star's RGB bound); table gradients are fp16 sums of fp16 products on both sides -- compared on
their norm and on the entries both sides touch."""
import numpy as np
import pytest
import torch

from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _pair(n_rays=1024, seed=3, small=True):
    from laenerf_amd import synthetic as S
    from laenerf_amd.network import NeRFNetwork
    from tools.reference_chain import ReferenceChain
    from laenerf_amd.renderer import NeRFRenderer
    torch.manual_seed(seed)
    chain = ReferenceChain(bound=1, min_near=0.2).to(DEV).train()
    net = NeRFNetwork(bound=1).to(DEV).train()
    r = NeRFRenderer(net, bound=1, min_near=0.2).to(DEV).train()
    with torch.no_grad():
        chain.embeddings.uniform_(-0.5, 0.5)                      # a table that matters to the image
        net.encoder.embeddings.copy_(chain.embeddings)
        net.sigma_net.weights.copy_(chain.sigma_net.weights)
        net.color_net.weights.copy_(chain.color_net.weights)
    bits = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(DEV)
    chain.density_bitfield = bits
    r.density_bitfield = bits.clone()
    o, d = S.lego_like_rays(n_rays, seed=seed, n_views=1)
    return chain, r, torch.from_numpy(o).to(DEV), torch.from_numpy(d).to(DEV), torch.rand(n_rays, 3, device=DEV)


def test_drop_in_forward_equals_fused_driver():
    chain, r, o, d, gt = _pair()
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        a = chain.render_train(o, d, perturb=False)
        b = r.render_train(o, d, bg_color=1, perturb=False)
    assert a["image"].shape == b["image"].shape == (o.shape[0], 3)
    assert float((a["image"] - b["image"]).abs().max()) <= 1e-4
    assert float((a["weights_sum"] - b["weights_sum"]).abs().max()) <= 1e-4
    hit = a["weights_sum"] > 0
    assert hit.float().mean() > 0.2                                # the scene is in view
    da, db = a["depth"][hit], b["depth"][hit]
    assert float((da - db).abs().max()) <= 1e-4


def test_drop_in_backward_matches_fused_driver():
    """gradients of the MSE against the same targets: MLP weights and hash table"""
    chain, r, o, d, gt = _pair()
    # perturb off on both sides: the two drivers draw their torch.rand noise at different points of the generator's stream
    with torch.autocast("cuda", dtype=torch.float16):
        oa = chain.render_train(o, d, perturb=False)
        la = torch.nn.functional.mse_loss(oa["image"], gt, reduction="none").mean(-1).mean()
        ob = r.render_train(o, d, bg_color=1, perturb=False)
        lb = torch.nn.functional.mse_loss(ob["image"], gt)
    (la * 1024).backward()
    (lb * 1024).backward()
    assert abs(float(la) - float(lb)) <= 1e-5
    net = r.model
    for pa, pb, name in ((chain.sigma_net.weights, net.sigma_net.weights, "sigma"), (chain.color_net.weights, net.color_net.weights, "colour")):
        ga, gb = pa.grad.float(), pb.grad.float()
        assert torch.isfinite(ga).all() and torch.isfinite(gb).all()
        rel = float((ga - gb).norm() / gb.norm().clamp_min(1e-20))
        assert rel < 2e-2, (name, rel)                             # fp16 inputs to the dW products, different summation orders
    ga, gb = chain.embeddings.grad.float(), net.encoder.embeddings.grad.float()
    assert float(gb.abs().max()) > 0
    rel = float((ga - gb).norm() / gb.norm())
    assert rel < 2e-2, rel
    assert torch.equal(ga != 0, gb != 0) or float(((ga != 0) ^ (gb != 0)).float().mean()) < 1e-3


def test_drop_in_step_runs_the_reference_sequence_and_learns():
    """a few optimizer steps through scaler.scale / backward / step / update: finite, sized by mean_count after 16 steps, loss falls"""
    from tools.reference_chain import drop_in_train_step
    chain, r, o, d, gt = _pair(n_rays=512)
    gt = torch.full_like(gt, 0.25)
    opt = torch.optim.Adam(chain.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    losses = []
    for i in range(20):
        loss, out = drop_in_train_step(chain, opt, scaler, (o, d, gt))
        losses.append(float(loss))
        if (i + 1) % 16 == 0:
            chain.update_mean_count()
    assert chain.mean_count > 0 and out["n_rows"] == chain.mean_count + 128 - chain.mean_count % 128
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
