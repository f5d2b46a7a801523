#!/usr/bin/env python3
"""host cost of one backend call (Python checks + ctypes + launch), microseconds: what an eager, operator-by-operator caller
(the reference's loops, the drop-in step) pays per operator on top of the kernel.  python tools/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_call(fn, n=3000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


def main():
    from laenerf_amd import _lib
    from laenerf_amd.backend import raymarching_backend as R, shencoder_backend as SH, ffmlp_backend as F
    dev = torch.device("cuda", 0)
    N = 64
    o = torch.rand(N, 3, device=dev); d = torch.nn.functional.normalize(torch.rand(N, 3, device=dev), dim=-1)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device=dev)
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    out = torch.empty(N, 16, device=dev)
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    po, pd, pa, pn, pf = o.data_ptr(), d.data_ptr(), aabb.data_ptr(), nears.data_ptr(), fars.data_ptr()
    x = torch.rand(256, 32, device=dev).half(); w = torch.rand(64 * (32 + 64 + 16), device=dev).half(); y = torch.empty(256, 16, device=dev, dtype=torch.half)
    res = {
        "torch_add (reference point)": per_call(lambda: torch.add(nears, 1.0, out=fars)),
        "raw ctypes call of lae_near_far_from_aabb": per_call(lambda: lib.lae_near_far_from_aabb(po, pd, pa, N, 0.2, pn, pf, s)),
        "backend near_far_from_aabb": per_call(lambda: R.near_far_from_aabb(o, d, aabb, N, 0.2, nears, fars)),
        "backend sh_encode_forward": per_call(lambda: SH.sh_encode_forward(d, out, N, 3, 4, None)),
        "backend ffmlp_inference": per_call(lambda: F.ffmlp_inference(x, w, 256, 32, 16, 64, 2, 0, 6, None, y)),
        "torch.cuda.current_stream().cuda_stream": per_call(lambda: torch.cuda.current_stream().cuda_stream),
    }
    for k, v in res.items():
        print(f"{v:8.2f} us  {k}")


if __name__ == "__main__":
    main()
