"""debug: fused recompute backward against the oracle's backward fed with the KERNEL's forward activations (mode-1 forward buffer)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gpu_util import DEV, N, half_from_bits, bits_from_half
from oracle import oracle as O
from laenerf_amd.backend import ffmlp_backend as F
O.build()
for c in (dict(IN=64, H=64, NL=2, tiles=38, seed=50), dict(IN=48, H=64, NL=3, tiles=6, seed=2515)):
    IN, H, NL, B = c["IN"], c["H"], c["NL"], 16 * c["tiles"]
    rng = np.random.default_rng(c["seed"])
    nW = O.ffmlp_num_params(IN, H, NL)
    Wh = O.to_f16_bits(rng.uniform(-np.sqrt(3 / H), np.sqrt(3 / H), nW).astype(np.float32))
    Xh = O.to_f16_bits(rng.uniform(-1, 1, (B, IN)).astype(np.float32))
    ref_out, ref_fb = O.ffmlp_forward(Xh, Wh, IN, 16, H, NL, activation=0)
    Gh = O.to_f16_bits((rng.standard_normal((B, 16)) * 0.05).astype(np.float32))
    F.ffmlp_set_mode(1)
    out = torch.empty(B, 16, device=DEV, dtype=torch.half); fb = torch.empty(NL, B, H, device=DEV, dtype=torch.half)
    F.ffmlp_forward(half_from_bits(Xh), half_from_bits(Wh), B, IN, 16, H, NL, 0, 6, fb, out)
    F.ffmlp_set_mode(0)
    fbk = bits_from_half(fb)
    print(c, "forward buffer: kernel vs oracle differing entries", int((fbk != ref_fb).sum()), "mask differences", int(((fbk & 0x7fff) == 0).__xor__((ref_fb & 0x7fff) == 0).sum()))
    for name, buf in (("oracle fwd buffer", ref_fb), ("kernel fwd buffer", fbk)):
        ref_gw, ref_gi, _ = O.ffmlp_backward(Gh, Xh, Wh, buf, IN, 16, H, NL, calc_grad_inputs=True)
        gi = torch.zeros(B, IN, device=DEV, dtype=torch.half); gw = torch.zeros(nW, device=DEV, dtype=torch.half)
        F.ffmlp_backward(half_from_bits(Gh), half_from_bits(Xh), half_from_bits(Wh), None, B, IN, 16, H, NL, 0, 6, True, None, gi, gw)
        a, b = N(gi).astype(np.float64), O.from_f16_bits(ref_gi).astype(np.float64)
        a2, b2 = N(gw).astype(np.float64), O.from_f16_bits(ref_gw).astype(np.float64)
        print(f"   vs oracle backward on the {name}: gi max err {np.abs(a-b).max():.3e} (max {np.abs(b).max():.2e}), gw max err {np.abs(a2-b2).max():.3e} (max {np.abs(b2).max():.2e})")
