"""teacher -> student scene fitting on the HIP path: a NeRFNetwork with random (but structured) parameters renders the
target colours, a fresh network is trained on them with the reference's loop shape (update_extra_state every 16 steps,
Adam lr 1e-2 with the 0.1^(it/iters) decay, GradScaler).  Prints PSNR on held-out rays.  python tools/fit_scene.py [steps]

PSNR A/B (north_star "PSNR within 0.05 dB of reference", PSNRMeter nerf/utils.py:222): python tools/fit_scene.py --ab [steps] [seeds]
fits the SAME teacher with the same ray batches and student seeds three ways:
  fused       this repository's default path: fused field op, fused criterion, FusedAdam (GradScaler inside), fp16 shadow table
  operators   the reference-shaped op sequence on the same architecture (network_ff.py): grid_encode -> ffmlp -> trunc_exp ->
              sh_encode -> ffmlp -> sigmoid -> composite_rays_train, torch mse_loss, torch.optim.Adam + torch.amp.GradScaler
  linear      the reference's default nets (nerf/network.py:95-124, NeRFNetworkLinear: nn.Linear chains under autocast), same ops
              and optimizer as `operators` -- another architecture (one hidden GEMM less in the sigma net), reported beside"""
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


def fit(steps=400, n_rays=4096, dev=torch.device("cuda:0"), log=print, bound=1, opacity=1.5, lr=1e-2, variant="fused", seed=0):
    """bound = 2: two cascades (the forward-facing / unbounded configs of the reference, e.g. configs_llff/flower.sh).
    variant: see the module docstring; seed: the student's initialisation (the ray batches do not depend on it)"""
    teacher, tr = make_teacher(dev, bound, opacity)
    radius = 3.2 if bound == 1 else 2.6
    if variant != "fused":
        return _fit_reference_shaped(steps, n_rays, dev, log, bound, lr, variant, seed, tr, radius)

    def target(o, d):
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            return tr.render_eval(o, d, bg_color=1, max_steps=1024)["image"].float()
    torch.manual_seed(seed)
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


def _fit_reference_shaped(steps, n_rays, dev, log, bound, lr, variant, seed, tr, radius):
    """the reference's training step shape (Trainer.train_step + train_one_epoch, nerf/utils.py:1455-1500): autocast forward through
    the operator-by-operator network, `loss = criterion(pred, gt).mean()`, scaler.scale(loss).backward(), scaler.step(Adam),
    scaler.update(); Adam(betas=(0.9, 0.99), eps=1e-15) as main_nerf.py:223"""
    from laenerf_amd.network import NeRFNetworkLinear

    def target(o, d):
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            return tr.render_eval(o, d, bg_color=1, max_steps=1024)["image"].float()
    torch.manual_seed(seed)
    if variant == "linear":
        net = NeRFNetworkLinear(bound=bound).to(dev)
    else:
        net = NeRFNetwork(bound=bound).to(dev)
        net.fused_head = False; net.fused_field = False           # network_ff.py:51-81 operator by operator
    r = NeRFRenderer(net, bound=bound, density_thresh=10).to(dev)
    opt = torch.optim.Adam(net.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    net.train()
    t0 = time.perf_counter()
    for it in range(steps):
        if it % 16 == 0:
            with torch.autocast("cuda", dtype=torch.float16):
                r.update_extra_state()
        o, d = S.lego_like_rays(n_rays, seed=1000 + it, radius=radius)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        gt = target(o, d)
        for g in opt.param_groups:
            g["lr"] = lr * 0.1 ** min(it / steps, 1.0)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            res = r.render_train(o, d, bg_color=1, perturb=True)
            loss = torch.nn.functional.mse_loss(res["image"].float(), gt)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        if it % 100 == 0 or it == steps - 1:
            log(f"step {it:4d} loss {float(loss.detach()):.5f} samples {res['n_samples']} mean_count {r.mean_count} "
                f"occupied {float((r.density_grid > min(r.mean_density, r.density_thresh)).float().mean()):.3f}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    net.eval()
    o, d = S.lego_like_rays(16384, seed=7, radius=radius)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    gt = target(o, d)
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        pred = r.render_eval(o, d, bg_color=1, max_steps=1024, frame_loop=False)["image"].float()
    psnr = -10 * math.log10(float(((pred - gt) ** 2).mean()))
    log(f"held-out PSNR {psnr:.2f} dB after {steps} steps ({dt:.1f} s incl. target renders), variant {variant}, loss scale {scaler.get_scale():.0f}")
    return psnr


def psnr_ab(steps=400, seeds=(0, 1, 2), log=print, bound=1):
    """-> {variant: [PSNR per seed]}; same teacher, same ray batches, same student seeds for the three variants"""
    out = {}
    for variant in ("fused", "operators", "linear"):
        out[variant] = [fit(steps=steps, log=lambda *_: None, bound=bound, variant=variant, seed=sd) for sd in seeds]
        log(f"{variant:10s} PSNR per seed {[round(p, 2) for p in out[variant]]}  mean {np.mean(out[variant]):.3f} dB  spread {np.ptp(out[variant]):.3f} dB")
    log(f"fused - operators: {np.mean(out['fused']) - np.mean(out['operators']):+.3f} dB (same architecture)   "
        f"fused - linear: {np.mean(out['fused']) - np.mean(out['linear']):+.3f} dB (the reference's default nets, another architecture)")
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--ab":
        psnr_ab(int(sys.argv[2]) if len(sys.argv) > 2 else 400, tuple(range(int(sys.argv[3]) if len(sys.argv) > 3 else 3)))
        sys.exit(0)
    fit(int(sys.argv[1]) if len(sys.argv) > 1 else 400, bound=int(sys.argv[2]) if len(sys.argv) > 2 else 1,
        opacity=float(sys.argv[3]) if len(sys.argv) > 3 else 1.5, lr=float(sys.argv[4]) if len(sys.argv) > 4 else 1e-2)
