"""CPU: the measurement harness of the zero-edit drop-in step (tools/reference_chain.py) follows the allocate-then-call rules of
the reference's operator wrappers as the fixtures recorded them from the reference's own Python (tests/golden/ops_wrappers.npz,
ffmlp_init.npz: make_golden.py runs raymarching/raymarching.py and ffmlp/ffmlp.py unmodified over recording backends).

The four backend modules are replaced by recorders under their reference names (`_raymarching`, `_gridencoder`, `_shencoder`, `_ffmlp`)
-- exactly how the reference's wrappers find a backend -- so nothing here needs a GPU.  Checked: the sample-buffer size rule
(`mean_count` rounded up by a FULL align when already aligned, host-sized while mean_count <= 0), zero-filled sample buffers, the
counter ring, the FFMLP row padding that is always added, forward / backward buffer shapes and the zero fill of the latter, the
`[L,B,C]` encoder output and level-major gradient, the zero-filled table gradient, the FFMLP parameter vector (seed-42 init), the
argument lists of every backend call (raymarching/src/bindings.cpp:5-20, gridencoder :5-8, shencoder :5-7, ffmlp :5-10).
(The half-precision casts of the wrappers need CUDA autocast and are exercised by tests/test_gpu_dropin.py.)"""
import sys
import types

import numpy as np
import pytest
import torch

from conftest import golden


class Recorder:
    def __init__(self, g):
        self.g, self.calls = g, []

    def module(self, name, fns):
        m = types.ModuleType(name)
        for f in fns:
            setattr(m, f, getattr(self, f))
        return m

    # ---- _raymarching
    def near_far_from_aabb(self, rays_o, rays_d, aabb, N, min_near, nears, fars):
        self.calls.append(("near_far", N, float(min_near), tuple(aabb.tolist())))
        nears.copy_(torch.from_numpy(self.g["nears"][:N])); fars.copy_(torch.from_numpy(self.g["fars"][:N]))

    def march_rays_train(self, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
        assert xyzs.shape == (M, 3) and dirs.shape == (M, 3) and deltas.shape == (M, 2) and rays.shape == (N, 3) and rays.dtype == torch.int32
        assert not xyzs.any() and not dirs.any() and not deltas.any()          # torch.zeros: the reference's kernel relies on it (raymarching.py:207-209)
        assert counter.shape == (2,) and counter.dtype == torch.int32 and int(counter[0]) == 0 and noises.shape == (N,)
        self.calls.append(("march", N, M, int(C), int(H), float(bound), float(dt_gamma), int(max_steps), bool((noises > 0).any())))
        counter[0] = int(self.g["counter"][0]); counter[1] = N
        rays.copy_(torch.from_numpy(self.g["rays"][:N]))

    def composite_rays_train_forward(self, sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        assert sigmas.shape == (M,) and rgbs.shape == (M, 3) and weights_sum.shape == (N,) and image.shape == (N, 3)
        self.calls.append(("composite_fwd", M, N, float(T_thresh)))
        weights_sum.fill_(0.5); depth.fill_(1.0); image.fill_(0.25)

    def composite_rays_train_backward(self, g_ws, g_img, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, g_sigmas, g_rgbs):
        assert g_sigmas.shape == (M,) and g_rgbs.shape == (M, 3) and not g_sigmas.any() and not g_rgbs.any()     # zeros_like (raymarching.py:283-284)
        self.calls.append(("composite_bwd", M, N))
        g_sigmas.fill_(1e-3); g_rgbs.fill_(1e-3)

    # ---- _gridencoder
    def grid_encode_forward(self, inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
        assert outputs.shape == (L, B, C) and outputs.is_contiguous() and inputs.shape == (B, D) and dy_dx is None
        assert float(inputs.min()) >= 0.0 and float(inputs.max()) <= 1.0       # (x + bound) / (2 bound) happened in Python (grid.py:149)
        self.calls.append(("grid_fwd", B, D, C, L, round(float(S), 6), int(H), gridtype, bool(align_corners), interp, tuple(embeddings.shape)))
        outputs.normal_(0, 0.1)

    def grid_encode_backward(self, grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
        assert grad.shape == (L, B, C) and grad.is_contiguous()                # [B, L*C] -> [L, B, C] copy (grid.py:75)
        assert grad_embeddings.shape == embeddings.shape and not grad_embeddings.any() and grad_inputs is None
        self.calls.append(("grid_bwd", B, L, C))
        grad_embeddings.fill_(1e-4)

    # ---- _shencoder
    def sh_encode_forward(self, inputs, outputs, B, D, C, dy_dx):
        assert outputs.shape == (B, C * C) and inputs.dtype == torch.float32 and dy_dx is None
        self.calls.append(("sh_fwd", B, D, C))
        outputs.normal_(0, 0.3)

    # ---- _ffmlp
    def allocate_splitk(self, n):
        self.calls.append(("splitk", int(n)))

    def ffmlp_forward(self, inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer, outputs):
        assert inputs.shape == (B, input_dim) and outputs.shape == (B, output_dim) and forward_buffer.shape == (num_layers, B, hidden_dim)
        assert B % 128 == 0 and not inputs[-1].any()                            # at least one appended row, and it is zeros (ffmlp.py:157-159)
        self.calls.append(("ffmlp_fwd", B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, weights.numel()))
        outputs.normal_(0, 0.1)

    def ffmlp_backward(self, grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                       calc_grad_inputs, backward_buffer, grad_inputs, grad_weights):
        assert grad.shape == (B, output_dim) and backward_buffer.shape == (num_layers, B, hidden_dim) and not backward_buffer.any()   # torch.zeros (ffmlp.py:71)
        assert grad_weights.shape == weights.shape and not grad_weights.any() and grad_inputs.shape == inputs.shape and calc_grad_inputs
        self.calls.append(("ffmlp_bwd", B, input_dim, hidden_dim, num_layers))
        grad_inputs.fill_(1e-3); grad_weights.fill_(1e-3)


@pytest.fixture
def chain_on_recorders(monkeypatch):
    g = golden("ops_wrappers")
    rec = Recorder(g)
    mods = {"_raymarching": ("near_far_from_aabb", "march_rays_train", "composite_rays_train_forward", "composite_rays_train_backward"),
            "_gridencoder": ("grid_encode_forward", "grid_encode_backward"), "_shencoder": ("sh_encode_forward",),
            "_ffmlp": ("allocate_splitk", "ffmlp_forward", "ffmlp_backward")}
    for name, fns in mods.items():
        monkeypatch.setitem(sys.modules, name, rec.module(name, fns))
    from tools.reference_chain import ReferenceChain
    chain = ReferenceChain(bound=1, min_near=0.2).train()
    chain.density_bitfield = torch.from_numpy(g["bitfield"])
    return chain, rec, g


def test_chain_follows_the_wrapper_rules_the_fixture_recorded(chain_on_recorders):
    chain, rec, g = chain_on_recorders
    fi = golden("ffmlp_init")
    # FFMLP.__init__: flat fp32 parameter vector, seed-42 uniform init, allocate_splitk(num_layers + 1) (ffmlp.py:118-144)
    for net, name, layers in ((chain.sigma_net, "sigma", 2), (chain.color_net, "color", 3)):
        assert net.weights.numel() == int(fi[name + "_n"]) and net.padded_output_dim == int(fi[name + "_padded_out"])
        assert np.array_equal(net.weights.detach().numpy()[:256], fi[name + "_head"])
        assert ("splitk", layers + 1) in rec.calls
    assert tuple(chain.embeddings.shape) == (6119864, 2) and chain.cascade == 1
    o, d = torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"])
    gt = torch.full((128, 3), 0.5)
    total = int(g["counter"][0])
    rec.calls.clear()
    # host-sized march (mean_count <= 0): M = N * max_steps, trimmed to the counter rounded up by a full align (raymarching.py:223-231)
    loss, out = chain.train_loss(o, d, gt)
    march = [c for c in rec.calls if c[0] == "march"][0]
    assert march[1:3] == (128, 128 * 1024) and march[3:8] == (1, 128, 1.0, 0.0, 1024) and march[8] is True      # perturb=True: torch.rand noise
    assert out["n_rows"] == int(g["M_trimmed"]) == total + (128 - total % 128)
    assert int(chain.step_counter[0, 0]) == total and chain.local_step == 1
    Mrows = out["n_rows"]
    # the network: encoder on [Mrows] rows ([L,B,C] out), both FFMLPs on Mrows + 128 rows (always padded), SH degree 4
    grid = [c for c in rec.calls if c[0] == "grid_fwd"][0]
    assert grid[1:6] == (Mrows, 3, 2, 16, round(float(np.log2(chain.per_level_scale)), 6)) and grid[6:10] == (16, 0, False, 0)
    ff = [c for c in rec.calls if c[0] == "ffmlp_fwd"]
    Bp = Mrows + (128 - Mrows % 128)
    assert ff[0][1:8] == (Bp, 32, 16, 64, 2, 0, 6) and ff[1][1:8] == (Bp, 32, 16, 64, 3, 0, 6) and Bp == Mrows + 128   # Mrows is a multiple of 128: a full tile of padding
    assert [c for c in rec.calls if c[0] == "sh_fwd"][0][1:] == (Mrows, 3, 4)
    assert [c for c in rec.calls if c[0] == "composite_fwd"][0][1:] == (Mrows, 128, pytest.approx(1e-4))
    assert [c for c in rec.calls if c[0] == "near_far"][0][1:3] == (128, pytest.approx(0.2))
    # backward through the recorders: level-major encoder gradient, zero-filled table gradient / backward buffers / compositing gradients
    rec.calls.clear()
    loss.backward()
    kinds = [c[0] for c in rec.calls]
    assert kinds == ["composite_bwd", "ffmlp_bwd", "ffmlp_bwd", "grid_bwd"], kinds
    assert rec.calls[1][1:] == (Bp, 32, 64, 3) and rec.calls[2][1:] == (Bp, 32, 64, 2) and rec.calls[3][1:] == (Mrows, 16, 2)
    assert chain.embeddings.grad is not None and chain.embeddings.grad.shape == chain.embeddings.shape
    assert chain.sigma_net.weights.grad is not None and chain.color_net.weights.grad is not None
    # mean_count mode: M = mean_count + (align - mean_count % align), a full align more when already aligned (raymarching.py:201-204)
    for mc in (1000, 1024, 5000):
        chain.mean_count = mc
        rec.calls.clear()
        with torch.no_grad():
            res = chain.render_train(o, d)
        assert [c for c in rec.calls if c[0] == "march"][0][2] == int(g[f"mc{mc}_M"]) == res["n_rows"]
    # update_extra_state's mean_count refresh from the counter ring (renderer.py:644-647)
    chain.update_mean_count()
    assert chain.mean_count == total and chain.local_step == 0


@pytest.mark.parametrize("B", [1, 127, 128, 129, 256])
def test_ffmlp_padding_rule_of_the_chain(chain_on_recorders, B):
    chain, rec, g = chain_on_recorders
    fi = golden("ffmlp_init")
    want = int(fi["pad_B_backend"][list(fi["pad_B_in"]).index(B)])
    rec.calls.clear()
    y = chain.sigma_net(torch.rand(B, 32))
    assert y.shape == (B, 16) and [c for c in rec.calls if c[0] == "ffmlp_fwd"][0][1] == want
