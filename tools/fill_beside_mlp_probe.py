#!/usr/bin/env python3
"""VERDICT r5 item 4, priced before anything is built: would the fill pass of the hash-grid backward (k_bwd_walk<FILL>, 38 %
SQ_ACTIVE_INST: instruction-bound, 3 waves per SIMD) overlap the MFMA-issue-bound MLP backward (k_mlp_bwd_wave, one wave per SIMD)
if the two shared the chip -- which is what fusing the fill into the producer of its gradients would buy at best?

Needs a probe build of the library (the fill and accumulate passes launchable alone):
    python -m laenerf_amd.build --out tools/ubench/bin/liblaenerf_phase.so -DLAE_GRID_BWD_PHASE_PROBE
    LAE_HIP_LIB=tools/ubench/bin/liblaenerf_phase.so python tools/fill_beside_mlp_probe.py
Measures, with HIP events behind a spin kernel, on the headline's sample set (4096 lego-like rays, ~250 k samples): the fused head
backward alone; the fill pass alone; the accumulate pass alone; fill on a second stream BESIDE the head backward; and the same pair
with the fill working on HALF the samples (the verdict's form).  Prints one JSON line.  Close the idea unless pair < 0.85 x sum."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from laenerf_amd import synthetic as S
    from laenerf_amd.backend import ffmlp_backend as F, gridencoder_backend as G
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    from laenerf_amd.streams import concurrent_side_stream
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = NeRFNetwork(bound=1).to(dev)
    r = NeRFRenderer(net, bound=1, min_near=0.2).to(dev)
    r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
    o, d = S.lego_like_rays(4096, seed=0, n_views=1)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    from laenerf_amd import raymarching as RM
    nears, fars = RM.near_far_from_aabb(o, d, r.aabb_train, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = RM.march_rays_train(o, d, 1.0, r.density_bitfield, 1, 128, nears, fars, counter, -1, True, 128, False, 0, 1024)
    M = xyzs.shape[0] // 16 * 16
    xyzs, dirs = xyzs[:M].contiguous(), dirs[:M].contiguous()
    enc = net.encoder
    L, C = 16, 2
    S_ = float(np.log2(enc.per_level_scale))
    table = enc.embeddings.detach().half()
    ws, wc = net.sigma_net.weights.detach().half(), net.color_net.weights.detach().half()
    in_map = (1.0, 0.5)
    feats = torch.empty(M, L * C, device=dev, dtype=torch.half)
    G.grid_encode_forward(xyzs, table, enc.offsets, feats, M, 3, C, L, S_, 16, None, 0, False, 0, blc=True, in_map=in_map, offsets_host=enc.offsets_host)
    h = torch.empty(M, 16, device=dev, dtype=torch.half); sig = torch.empty(M, device=dev); rgb = torch.empty(M, 3, device=dev)
    F.nerf_head_forward(feats, dirs, ws, wc, M, 1.0, h, sig, rgb)
    g_sig, g_rgb = torch.randn(M, device=dev) * 1e-3, torch.randn(M, 3, device=dev) * 1e-3
    g_h, g_enc = torch.empty(M, 16, device=dev, dtype=torch.half), torch.empty(M, L * C, device=dev, dtype=torch.half)
    g_ws, g_wc = torch.zeros_like(ws), torch.zeros_like(wc)
    g_lm = (torch.randn(L, M, C, device=dev) * 1e-3).half()                 # level-major gradient for the planned backward
    g_table = torch.zeros_like(table)

    def head_bwd():
        F.nerf_head_backward(g_sig, g_rgb, feats, dirs, h, rgb, ws, wc, M, 1.0, g_h, g_enc, g_ws, g_wc)

    def make_grid(m):
        x = xyzs[:m].contiguous()
        g = g_lm[:, :m].contiguous()
        plan = G.grid_backward_plan(x, enc.offsets, m, 3, C, L, S_, 16, 0, False, 0, True, in_map=in_map, offsets_host=enc.offsets_host)

        def run():
            G.grid_encode_backward(g, x, table, enc.offsets, g_table, m, 3, C, L, S_, 16, None, None, 0, False, 0, in_map=in_map,
                                   offsets_host=enc.offsets_host, plan=plan)
        return run
    grid_full, grid_half = make_grid(M), make_grid(M // 32 * 16)
    main_s = torch.cuda.current_stream()
    side, probe = concurrent_side_stream()

    def timed(fa, fb=None, reps=20):
        """median us of fa on the main stream (and fb beside it on the side stream), events behind a spin"""
        out = []
        for _ in range(reps + 3):
            torch.cuda._sleep(3_000_000)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(main_s)
            if fb is not None:
                side.wait_event(e0)
            fa()
            e1.record(main_s)
            if fb is not None:
                with torch.cuda.stream(side):
                    fb()
                    e2.record(side)
            torch.cuda.synchronize()
            out.append(max(e0.elapsed_time(e1), e0.elapsed_time(e2) if fb is not None else 0.0) * 1e3)
        return float(np.median(out[3:]))

    res = {"samples": int(M), "side_stream": probe}
    os.environ["LAE_GRID_BWD_PHASE"] = "0"
    res["head_backward_us"] = timed(head_bwd)
    res["fill_plus_accumulate_us"] = timed(grid_full)
    os.environ["LAE_GRID_BWD_PHASE"] = "2"
    res["accumulate_alone_us"] = timed(grid_full)
    os.environ["LAE_GRID_BWD_PHASE"] = "1"
    res["fill_alone_us"] = timed(grid_full)
    res["fill_half_alone_us"] = timed(grid_half)
    res["pair_head_backward_and_fill_us"] = timed(head_bwd, grid_full)
    res["pair_head_backward_and_fill_half_us"] = timed(head_bwd, grid_half)
    for k, fk in (("full", "fill_alone_us"), ("half", "fill_half_alone_us")):
        pair = res["pair_head_backward_and_fill_us" if k == "full" else "pair_head_backward_and_fill_half_us"]
        res[f"pair_over_sum_{k}"] = round(pair / (res["head_backward_us"] + res[fk]), 3)
    res = {k: (round(v, 1) if isinstance(v, float) and "over" not in k else v) for k, v in res.items()}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
