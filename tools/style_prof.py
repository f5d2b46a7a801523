"""profile target: LAENeRF palette-network steps (the step of bench.py style_step), eager (rocprofv3 --kernel-trace --stats -- python tools/style_prof.py)"""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
from laenerf_amd.editing import LAENeRF
from laenerf_amd.optim import FusedAdam
dev = torch.device("cuda:0")
P = 100000
params = SimpleNamespace(bound=1, num_palette_bases=8, style_weight=0, weight_loss_uniform=1e-3, weight_loss_non_uniform=1e-3,
                         offset_loss=1e-2, palette_loss_valid=1.0, palette_loss_distinct=1e-2)
torch.manual_seed(7)
m = LAENeRF(params, dir_encoding="sphere_harmonics").to(dev).train()
opt = FusedAdam(m, param_groups=m.get_params(1e-3), betas=(0.9, 0.999), eps=1e-8)
v = torch.randn(P, 3, device=dev)
x = v / v.norm(dim=-1, keepdim=True) * 0.3 * torch.rand(P, 1, device=dev) ** (1 / 3)
d = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
target = torch.rand(P, 3, device=dev)
for it in range(12):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.float16):      # the step bench.py's style_step captures (fused point losses)
        loss, pred, w, o = m.forward_train_loss(x, d, target, params, opt, with_palet_loss=True)
    opt.backward(loss)
    opt.step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 10 * 1e3)
