#!/usr/bin/env python3
"""K eager steps of the zero-edit drop-in train step (tools/reference_chain.py) after its warm-up, for rocprofv3:

    rocprofv3 --kernel-trace --stats -d gpurun_out/dropin_k10 -- python3 tools/dropin_profile.py --steps 10
    rocprofv3 --kernel-trace --stats -d gpurun_out/dropin_k30 -- python3 tools/dropin_profile.py --steps 30

tools/dropin_launches.py takes the difference of the two kernel tables (calls and device time of 20 steady steps).
--torch-profiler: count kernels with torch.profiler instead (prints a table; no rocprofv3 around it)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-nan-check", action="store_true")
    ap.add_argument("--torch-profiler", action="store_true")
    a = ap.parse_args()
    from laenerf_amd import build, synthetic as S
    from tools.reference_chain import ReferenceChain, drop_in_train_step
    build.build()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    chain = ReferenceChain(bound=1, min_near=0.2, nan_check=not a.no_nan_check).to(dev).train()
    chain.density_bitfield = torch.from_numpy(S.pack_bits_np(S.sphere_density_grid(), 10.0)).to(dev)
    opt = torch.optim.Adam(chain.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    batches = []
    for b in range(16):
        o, d = S.lego_like_rays(4096, seed=b, n_views=1)
        batches.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.rand(4096, 3, device=dev)))
    for i in range(34):
        drop_in_train_step(chain, opt, scaler, batches[i % 16])
        if (i + 1) % 16 == 0:
            chain.update_mean_count()

    def run():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(a.steps):
            drop_in_train_step(chain, opt, scaler, batches[i % 16])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3
    if a.torch_profiler:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            ms = run()
        ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        tot = sum(e.device_time for e in ev) if ev and hasattr(ev[0], "device_time") else sum(e.cuda_time for e in ev)
        print(json.dumps({"steps": a.steps, "wall_ms_per_step": round(ms, 4), "device_kernels_per_step": len(ev) / a.steps,
                          "device_us_per_step": tot / a.steps}))
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40))
    else:
        print(json.dumps({"steps": a.steps, "wall_ms_per_step": round(run(), 4), "mean_count": chain.mean_count}))


if __name__ == "__main__":
    main()
