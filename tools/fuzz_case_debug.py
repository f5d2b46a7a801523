"""debug helper: one fuzz case of tests/test_gpu_frame_fuzz.py across budgets / max_n_step (prints weights_sum, image[0])"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_gpu_frame_fuzz as F
from gpu_util import T
from laenerf_amd import synthetic as S
c = dict(n=1, bound=1, occ="random", p=0.5, amp=0.05, T_thresh=1e-4, perturb=True, max_steps=16, max_n_step=2, budget_k=3, tiled=False,
         stream=None, distill=0, seed=8654, bg="white")
for k, v in [a.split("=") for a in sys.argv[1:]]:
    c[k] = type(c[k])(eval(v)) if not isinstance(c[k], str) else v
r = F.model(c["bound"], c["amp"])
bits = F.occupancy(c["occ"], r.cascade, c["bound"], c["p"], c["seed"])
r.density_bitfield = T(bits)
o, d = S.lego_like_rays(c["n"], seed=c["seed"], radius=3.2 if c["bound"] == 1 else 2.6)
o, d = T(o), T(d)
with torch.autocast("cuda", dtype=torch.float16):
    for perturb in (False, True):
        for mns in (1, 2, 8):
            torch.manual_seed(c["seed"])
            a = r.render_eval(o, d, frame_loop=False, want_stats=True, bg_color=1, perturb=perturb, max_steps=c["max_steps"], T_thresh=c["T_thresh"], max_n_step=mns)
            print(f"perturb {perturb} max_n_step {mns} operator: ws {a['weights_sum'][:4].tolist()} stats {a['stats']}")
            for bk in (0, 1, 2, 3, 8):
                torch.manual_seed(c["seed"])
                b = r.render_eval(o, d, frame_loop=True, want_stats=True, bg_color=1, perturb=perturb, max_steps=c["max_steps"], T_thresh=c["T_thresh"], max_n_step=mns, row_budget=bk * c["n"])
                print(f"    frame budget {bk}N: ws {b['weights_sum'][:4].tolist()} stats {b['stats']}")
