"""debug helper: generic FFMLP backward, new kernel (mode 0) against the cooperative one (mode 3), per 16-row tile"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from laenerf_amd.backend import ffmlp_backend as F
dev = "cuda:0"
for IN, NL, B in ((48, 2, 32), (48, 2, 128), (48, 3, 64), (32, 2, 64), (64, 3, 64)):
    g = torch.Generator(device=dev).manual_seed(1)
    nW = 64 * (IN + 64 * (NL - 1) + 16)
    W = ((torch.rand(nW, device=dev, generator=g) * 2 - 1) * 0.2).half()
    X = (torch.rand(B, IN, device=dev, generator=g) * 2 - 1).half()
    G = (torch.randn(B, 16, device=dev, generator=g) * 0.05).half()
    res = {}
    for mode in (3, 0):
        F.ffmlp_set_mode(mode)
        gi = torch.zeros(B, IN, device=dev, dtype=torch.half); gw = torch.zeros(nW, device=dev, dtype=torch.half)
        F.ffmlp_backward(G, X, W, None, B, IN, 16, 64, NL, 0, 6, True, None, gi, gw)
        res[mode] = (gi.float().cpu().numpy(), gw.float().cpu().numpy())
    d = np.abs(res[3][0] - res[0][0]).reshape(B // 16, 16, IN).max(axis=(1, 2))
    print(IN, NL, B, "gi max diff per tile", np.round(d, 4), "gw diff", np.abs(res[3][1] - res[0][1]).max())
F.ffmlp_set_mode(0)
