#!/usr/bin/env python3
"""one product path per call, for `rocprofv3 --kernel-trace --stats -- python3 tools/paths_for_trace.py <path>` (which kernels --
the library's and torch's own -- a path launches; tools/torch_kernel_audit.py reads the resulting kernel_stats.csv files):

    headline   the fused train step of bench.py (20 steps)            frame     eval_frame + frame1080 incl. the eight shards
    sharded    render_frame_sharded through a one-rank RCCL group     extract   editing.extract_views (configs[4] extraction)
    style      the LAENeRF palette step                               gridupd   update_extra_state sweeps
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    which = sys.argv[1]
    import torch
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from laenerf_amd import build
    build.build()
    if which == "headline":
        sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-frame", "--no-style", "--no-cpu-baseline", "--no-dropin"]
        bench.main()
    elif which == "frame":
        bench.eval_frame(dev); bench.eval_frame(dev, density_scale=30.0); bench.frame1080(dev)
    elif which == "sharded":
        import socket
        import torch.distributed as dist
        from laenerf_amd import dist as D, synthetic as S
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        D.FORCE_COLLECTIVES = True
        net, r = bench.eval_model(dev, bound=2, seed=1234)
        D.broadcast_model_state(r, src=0)
        H, W = 1080, 1920
        o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)

        def render(ro, rd):
            with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
                return r.render_eval(ro, rd, bg_color=1, max_steps=1024)
        for _ in range(3):
            D.render_frame_sharded(render, o, d, 0, 1, image_hw=(H, W))
        torch.cuda.synchronize()
        dist.destroy_process_group()
    elif which == "extract":
        bench.edit_extract(dev, n_views=4)
    elif which == "style":
        bench.style_step(dev)
    elif which == "gridupd":
        bench.grid_update(dev)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
