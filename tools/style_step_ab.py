"""A/B of the round-5 style-step glue removal: `bench.style_step` with the one-node input assembly
(`LAENeRF.fused_inputs`) and the shadow-aware MLP backward (`LAENeRF.ffmlp_shadows`) on and off, three alternating runs each.
    python tools/style_step_ab.py  >  profiles/r5_style_step_ab.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

dev = torch.device("cuda:0")
cases = [("operator chain, autograd .grad   ", dict(fused_inputs=False, ffmlp_shadows=False)),
         ("operator chain, shadow-aware MLPs", dict(fused_inputs=False, ffmlp_shadows=True)),
         ("one-node inputs, autograd .grad  ", dict(fused_inputs=True, ffmlp_shadows=False)),
         ("one-node inputs, shadow-aware    ", dict(fused_inputs=True, ffmlp_shadows=True))]
res = {name: [] for name, _ in cases}
for rep in range(3):
    for name, sw in cases:
        r = bench.style_step(dev, steps=200, switches=sw)
        res[name].append((r["one_graph_per_step_ms"], r["ms_per_step"]))
print("style step (configs[4], 100 000 points, HIP-graph replay, 200 steps per run), ms per step, three alternating runs")
print("one graph per step:")
for name, _ in cases:
    print(f"  {name}  " + "  ".join(f"{v[0]:.4f}" for v in res[name]) + f"   best {min(v[0] for v in res[name]):.4f}")
# the grouped scheme is quoted for the SHIPPED configuration only.  With the MLP gradients through autograd's .grad (ffmlp_shadows off)
# its replays ran at 1.0 instead of 0.33 ms per step whenever another style_step had run in the same process before (kernel times
# unchanged; 0.1-0.45 ms between consecutive main-graph replays in the kernel trace, growing with the number of replays) -- a
# combination nothing ships, not investigated further.
print("grouped two-stream scheme (counting half of the grid backward two groups ahead on a side stream; shipped configuration):")
for name, sw in cases:
    if sw["fused_inputs"] and sw["ffmlp_shadows"]:
        print(f"  {name}  " + "  ".join(f"{v[1]:.4f}" for v in res[name]) + f"   best {min(v[1] for v in res[name]):.4f}")
