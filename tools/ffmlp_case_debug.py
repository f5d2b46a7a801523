"""debug: one case of tests/test_gpu_fuzz_operators.py::test_fuzz_ffmlp_forward_backward -- error statistics of grad_inputs / grad_weights"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gpu_util import DEV, N, half_from_bits
from oracle import oracle as O
from laenerf_amd.backend import ffmlp_backend as F
O.build()
c = dict(IN=64, H=64, NL=2, tiles=38, act=0, seed=50)
for a in sys.argv[1:]:
    k, v = a.split("="); c[k] = int(v)
IN, H, NL, B = c["IN"], c["H"], c["NL"], 16 * c["tiles"]
rng = np.random.default_rng(c["seed"])
nW = O.ffmlp_num_params(IN, H, NL)
Wh = O.to_f16_bits(rng.uniform(-np.sqrt(3 / H), np.sqrt(3 / H), nW).astype(np.float32))
Xh = O.to_f16_bits(rng.uniform(-1, 1, (B, IN)).astype(np.float32))
ref_out, ref_fb = O.ffmlp_forward(Xh, Wh, IN, 16, H, NL, activation=0)
Gh = O.to_f16_bits((rng.standard_normal((B, 16)) * 0.05).astype(np.float32))
ref_gw, ref_gi, _ = O.ffmlp_backward(Gh, Xh, Wh, ref_fb, IN, 16, H, NL, calc_grad_inputs=True)
for mode in (0, 1, 3):
    F.ffmlp_set_mode(mode)
    fused = mode != 1 and F.fused_backward_available(IN, H, NL, 0)
    gi = torch.zeros(B, IN, device=DEV, dtype=torch.half); gw = torch.zeros(nW, device=DEV, dtype=torch.half)
    bb = None if fused else torch.empty(NL, B, H, device=DEV, dtype=torch.half)
    F.ffmlp_backward(half_from_bits(Gh), half_from_bits(Xh), half_from_bits(Wh), None if fused else half_from_bits(ref_fb), B, IN, 16, H, NL, 0, 6, True, bb, gi, gw)
    a, b = N(gi).astype(np.float64), O.from_f16_bits(ref_gi).astype(np.float64)
    d = np.abs(a - b)
    bad = np.argwhere(d > 4e-3 * np.abs(b).max() + 5e-4)
    print(f"mode {mode} fused {fused}: max |gi - ref| {d.max():.3e} (max |ref| {np.abs(b).max():.3e}); entries beyond tolerance {len(bad)}; rows {sorted(set(bad[:,0]))[:20]}")
    if len(bad):
        r = bad[0][0]
        print("   row", r, "got", a[r, :8], "ref", b[r, :8])
    a, b = N(gw).astype(np.float64), O.from_f16_bits(ref_gw).astype(np.float64)
    print(f"      gw max err {np.abs(a-b).max():.3e} (max |ref| {np.abs(b).max():.3e})")
F.ffmlp_set_mode(0)
# hypothesis for a single deviating row: a ReLU mask that flips between the kernel's recomputed activations and the oracle's stored ones
W = O.from_f16_bits(Wh).astype(np.float64); X = O.from_f16_bits(Xh).astype(np.float64)
W0 = W[:H * IN].reshape(H, IN); W1 = W[H * IN:H * IN + H * H].reshape(H, H)
z1 = X @ W0.T
h1 = np.maximum(O.from_f16_bits(O.to_f16_bits(z1.astype(np.float32))).astype(np.float64), 0)
z2 = h1 @ W1.T
for r in (606, 0, 1):
    print(f"row {r}: smallest |pre-activation| layer 1 {np.abs(z1[r]).min():.3e}, layer 2 {np.abs(z2[r]).min():.3e}")
print("rows with a layer-2 pre-activation within 2e-5 of zero:", np.argwhere(np.abs(z2).min(1) < 2e-5).ravel()[:20], "layer 1:", np.argwhere(np.abs(z1).min(1) < 2e-5).ravel()[:20])
