#!/usr/bin/env python3
"""Does FusedAdam's update pass (k_apply_multi: 294 MB, HBM-bound at ~5.9 TB/s) run faster when the parameter and moment tensors
(147 MB of fp32) were READ just before -- i.e. would a prefetch of them into the 256 MB Infinity Cache, issued on a side stream beside
the latency-bound accumulate pass of the grid backward, pay?  Events around opt.step() after (a) a 512 MB memset (cold), (b) nothing
special (the previous step's own writes), (c) a read of p / m / v (torch.sum of each) right before.  python tools/adam_mall_probe.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = NeRFNetwork(bound=1).to(dev)
    opt = FusedAdam(net, param_groups=net.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    sh = net.encoder.shadow
    p = net.encoder.embeddings
    m, v = opt.items[0][1], opt.items[0][2]
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    if sh.touched_lines is not None:
        sh.touched_lines.fill_(-1)

    def grads():
        sh.grad_half.normal_(0, 1e-3)
        sh.unreported = False

    def timed(prep, reps=15):
        out = []
        for _ in range(reps + 3):
            grads()
            prep()
            torch.cuda._sleep(500_000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); opt.step(); e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(out[3:]))
    res = {"cold_after_512MB_memset_us": timed(lambda: big.zero_()),
           "as_is_us": timed(lambda: None),
           "after_reading_p_m_v_us": timed(lambda: (p.detach().sum(), m.sum(), v.sum())),
           "after_reading_p_m_v_and_grad_us": timed(lambda: (p.detach().sum(), m.sum(), v.sum(), sh.grad_half.float().sum()))}
    t = []
    for _ in range(10):
        torch.cuda._sleep(500_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); p.detach().sum(); m.sum(); v.sum(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1) * 1e3)
    res["the_three_reads_alone_us"] = float(np.median(t))
    print(json.dumps({k: round(x, 1) for k, x in res.items()}))


if __name__ == "__main__":
    main()
