"""fused NeRF head forward / backward alone on M random rows (event-timed, back-to-back launches), A/B over the kernel
variants behind lae_ffmlp_set_mode; checks that the variants give the same results.

    python tools/mlp_bench.py [M]          # default 257792 rows (the bench batch)
Run under rocprofv3 --kernel-trace --stats for per-kernel durations."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from laenerf_amd.backend import ffmlp_backend as F   # noqa: E402

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 257792
M -= M % 16
g = torch.Generator(device=dev).manual_seed(0)
level_major = True
enc = (torch.randn(16, M, 2, device=dev, generator=g) * 0.5).half()            # level-major features
dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=-1).contiguous()
ws = ((torch.rand(64 * (32 + 64 + 16), device=dev, generator=g) * 2 - 1) * 0.2165).half()
wc = ((torch.rand(64 * (32 + 128 + 16), device=dev, generator=g) * 2 - 1) * 0.2165).half()


_f = None
_b = None


def fwd(fresh=False):
    global _f
    if _f is None or fresh:
        _f = (torch.empty(M, 16, dtype=torch.half, device=dev), torch.empty(M, device=dev), torch.empty(M, 3, device=dev))
    h, sig, rgb = _f
    F.nerf_head_forward(enc, dirs, ws, wc, M, 1.0, h, sig, rgb, level_major=level_major)
    return h, sig, rgb


def bwd(h, rgb, gs, gr, fresh=False):
    global _b
    if _b is None or fresh:
        _b = (torch.empty(M, 16, dtype=torch.half, device=dev), torch.empty(16, M, 2, dtype=torch.half, device=dev),
              torch.zeros_like(ws), torch.zeros_like(wc))
    gh, genc, gws, gwc = _b
    F.nerf_head_backward(gs, gr, enc, dirs, h, rgb, ws, wc, M, 1.0, gh, genc, gws, gwc, accumulate=False, level_major=level_major)
    return gh, genc, gws, gwc


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fwd_modes = [0]                     # one forward kernel ships (round 4 removed modes 16-18)
bwd_modes = [int(x) for x in os.environ.get("MLP_BENCH_BWD", "3,0").split(",") if x]
ref = None
for mode in fwd_modes:
    F.ffmlp_set_mode(mode)
    out = fwd(fresh=True)
    torch.cuda.synchronize()
    t = timed(fwd)
    msg = ""
    if ref is None:
        ref = out
    else:
        msg = " | vs first: " + ", ".join(f"{n} {'same bits' if torch.equal(a, b) else 'max diff %.3g' % float((a.float() - b.float()).abs().max())}"
                                          for n, a, b in zip(("h", "sigma", "rgb"), out, ref))
    if ref is not out and not torch.equal(out[2], ref[2]):
        bad = (out[2] != ref[2]).any(dim=1).nonzero().flatten()
        print(f"   rgb rows that differ: {bad.numel()} of {M}; first {bad[:8].tolist()}; new {out[2][bad[:3]].tolist()} old {ref[2][bad[:3]].tolist()}")
    print(f"head forward  mode {mode}: {t:7.1f} us  ({M} rows, {36864 * M / t / 1e9:.3f} PFLOP/s = {36864 * M / t / 1e9 / 2.5 * 100:.1f} % of 2.5){msg}", flush=True)
h, sig, rgb = ref
gs = torch.randn(M, device=dev, generator=g) * 1e-3
gr = torch.randn(M, 3, device=dev, generator=g) * 1e-3
refb = None
for mode in bwd_modes:
    F.ffmlp_set_mode(mode)
    out = bwd(h, rgb, gs, gr, fresh=True)
    torch.cuda.synchronize()
    t = timed(lambda: bwd(h, rgb, gs, gr))
    msg = ""
    if refb is None:
        refb = out
    else:
        msg = " | vs first: " + ", ".join(
            f"{n} {'same bits' if torch.equal(a, b) else 'max diff %.3g (max |ref| %.3g)' % (float((a.float() - b.float()).abs().max()), float(b.float().abs().max()))}"
            for n, a, b in zip(("grad_h", "grad_enc", "gW_sigma", "gW_color"), out, refb))
    print(f"head backward mode {mode}: {t:7.1f} us  ({73728 * M / t / 1e9:.3f} PFLOP/s){msg}", flush=True)
F.ffmlp_set_mode(0)
