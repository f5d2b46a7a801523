"""measured image difference between the fp32-table path (what the e2e fixtures were generated with) and the fp16-table
path (autocast, what training / the frame loop run) on the e2e fixtures; printed for DESIGN.md section 2"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import golden
from gpu_util import DEV, N, T
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
for tag in ("b1", "b2"):
    g = golden("e2e_" + tag)
    bound = int(g["bound"])
    out = {}
    for mode in ("fp32_table", "fp16_table"):
        net = NeRFNetwork(bound=bound, num_levels=16, log2_hashmap_size=10).to(DEV)
        net.encoder.embeddings.data = T(g["table"]); net.sigma_net.weights.data = T(g["sigma_w"]); net.color_net.weights.data = T(g["color_w"])
        if mode == "fp32_table":
            enc_fwd = net.encoder.forward
            def fp32_encoder(x, bound=1, enc_fwd=enc_fwd):
                with torch.autocast("cuda", enabled=False):
                    return enc_fwd(x.float(), bound=bound)
            net.encoder.forward = fp32_encoder
        r = NeRFRenderer(net, bound=bound, min_near=0.2).to(DEV)
        r.density_bitfield = T(g["bitfield"])
        net.eval()
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            ev = r.render_eval(T(g["rays_o"]), T(g["rays_d"]), bg_color=1, max_steps=256)
        out[mode] = N(ev["image"])
        print(tag, mode, "max |image - reference capture| = %.3e" % np.abs(out[mode] - g["eval_image"]).max())
    print(tag, "fp16 table vs fp32 table: max |d image| = %.3e" % np.abs(out["fp16_table"] - out["fp32_table"]).max())
