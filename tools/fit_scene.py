"""teacher -> student scene fitting on the HIP path: a NeRFNetwork with random (but structured) parameters renders the
target colours, a fresh network is trained on them with the reference's loop shape (update_extra_state every 16 steps,
Adam lr 1e-2 with the 0.1^(it/iters) decay, GradScaler).  Prints PSNR on held-out rays.  python tools/fit_scene.py [steps]"""
import sys, os, time, math, torch, numpy as np
sys.path.insert(0, os.getcwd())
from laenerf_amd import synthetic as S
from laenerf_amd.network import NeRFNetwork
from laenerf_amd.renderer import NeRFRenderer
from laenerf_amd.optim import FusedAdam


def make_teacher(dev, bound=1, opacity=1.5):
    torch.manual_seed(11)
    net = NeRFNetwork(bound=bound).to(dev).eval()
    net.encoder.embeddings.data.uniform_(-1.0, 1.0)
    net.sigma_net.weights.data.mul_(opacity)
    r = NeRFRenderer(net, bound=bound, density_thresh=10).to(dev).eval()
    grid = S.sphere_density_grid(cascade=r.cascade, bound=float(bound))                              # geometry: sphere + boxes
    r.density_bitfield = torch.from_numpy(S.pack_bits_np(grid, 10.0)).to(dev)
    return net, r


def fit(steps=400, n_rays=4096, dev=torch.device("cuda:0"), log=print, bound=1, opacity=1.5, lr=1e-2):
    """bound = 2: two cascades (the forward-facing / unbounded configs of the reference, e.g. configs_llff/flower.sh)"""
    teacher, tr = make_teacher(dev, bound, opacity)
    radius = 3.2 if bound == 1 else 2.6

    def target(o, d):
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            return tr.render_eval(o, d, bg_color=1, max_steps=1024)["image"].float()
    torch.manual_seed(0)
    net = NeRFNetwork(bound=bound).to(dev)
    r = NeRFRenderer(net, bound=bound, density_thresh=10).to(dev)
    opt = FusedAdam(net, param_groups=net.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
    net.train()
    t0 = time.perf_counter()
    for it in range(steps):
        if it % 16 == 0:
            with torch.autocast("cuda", dtype=torch.float16):
                r.update_extra_state()
        o, d = S.lego_like_rays(n_rays, seed=1000 + it, radius=radius)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        gt = target(o, d)
        opt.set_lr(lr * 0.1 ** min(it / steps, 1.0))
        with torch.autocast("cuda", dtype=torch.float16):
            res = r.render_train(o, d, bg_color=1, perturb=True, gt=gt, scaler=opt)
        res["loss"].backward()
        opt.step()
        if it % 100 == 0 or it == steps - 1:
            log(f"step {it:4d} loss {float(res['loss'].unscaled):.5f} samples {res['n_samples']} mean_count {r.mean_count} "
                f"occupied {float((r.density_grid > min(r.mean_density, r.density_thresh)).float().mean()):.3f}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    net.eval()
    o, d = S.lego_like_rays(16384, seed=7, radius=radius)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    gt = target(o, d)
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        pred = r.render_eval(o, d, bg_color=1, max_steps=1024)["image"].float()
    mse = float(((pred - gt) ** 2).mean())
    psnr = -10 * math.log10(mse)
    log(f"held-out PSNR {psnr:.2f} dB after {steps} steps ({dt:.1f} s incl. target renders), steps skipped by the scaler: {opt.steps_skipped}")
    return psnr


if __name__ == "__main__":
    fit(int(sys.argv[1]) if len(sys.argv) > 1 else 400, bound=int(sys.argv[2]) if len(sys.argv) > 2 else 1,
        opacity=float(sys.argv[3]) if len(sys.argv) > 3 else 1.5, lr=float(sys.argv[4]) if len(sys.argv) > 4 else 1e-2)
