#!/usr/bin/env python3
"""Reproducer / probe driver for round 4's unexplained fault: with the row code of k_grid_fwd_lean inside a
by-reference lambda inside a loop, frames of the device-resident loop differed from run to run as soon as a SECOND PROCESS used
the GPU.  This process renders the SAME frame `--frames` times (the dist_check frame: bonsai-shaped model, bound 2, 256 x 192
rays, one sample per ray and iteration) and compares every frame's bits with the first; a neighbour process is started first:

    --neighbour none | same | alu | stream | mfma | lds    same: a second copy of this script (its own result is reported too);
                                                alu / stream / mfma / lds: tools/ubench/bin/spinner (pure VALU work / 1 GiB copies / back-to-back MFMA / LDS traffic)
    --lib PATH                                  the library under test (default: the shipped in-tree one);
                                                tools/grid_loop_fault.sh builds the probe variants into tools/ubench/bin/

Prints ONE JSON line: frames that differ from the first, rays that differ (max over frames), the run structure of the
differing rays of the first differing frame.  Children are started BEFORE this process touches the GPU and are ended by PID."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def runs_of(idx):
    runs, st, pv = [], idx[0], idx[0]
    for v in idx[1:]:
        if v > pv + 1:
            runs.append((st, pv - st + 1)); st = v
        pv = v
    runs.append((st, pv - st + 1))
    return runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--neighbour", choices=["none", "same", "alu", "stream", "mfma", "lds"], default="same")
    ap.add_argument("--as-neighbour", action="store_true")
    ap.add_argument("--operator-loop", action="store_true", help="the reference-style operator loop instead of lae_render_frame")
    ap.add_argument("--neighbour-lib", default=None, help="library of the `same` neighbour (default: --lib)")
    ap.add_argument("--table-scale", type=float, default=0.5, help="hash table ~ U(-s, s)")
    ap.add_argument("--neighbour-table-scale", type=float, default=None, help="the neighbour's table scale (default: --table-scale): "
                    "a neighbour with very different feature values tells a cross-process leak from a private fault")
    ap.add_argument("--shift-va-mb", type=int, default=0, help="allocate this many MB first, so that this process's buffers sit at "
                    "other virtual addresses than the neighbour's")
    args = ap.parse_args()
    if args.lib:
        os.environ["LAE_HIP_LIB"] = os.path.abspath(args.lib)
    child = None
    if not args.as_neighbour and args.neighbour != "none":
        if args.neighbour == "same":
            nlib = args.lib if args.neighbour_lib is None else args.neighbour_lib
            nscale = args.table_scale if args.neighbour_table_scale is None else args.neighbour_table_scale
            cmd = [sys.executable, os.path.abspath(__file__), "--as-neighbour", "--frames", str(args.frames * 3), "--table-scale", str(nscale)] \
                + (["--lib", nlib] if nlib else []) + (["--operator-loop"] if args.operator_loop else [])
        else:
            cmd = [os.path.join(ROOT, "tools", "ubench", "bin", "spinner"), args.neighbour, "600"]
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        line = child.stdout.readline()                        # "ready": the neighbour is on the GPU
        assert "ready" in line, line

    import torch
    from laenerf_amd import synthetic as S
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.renderer import NeRFRenderer
    dev = torch.device("cuda", 0)
    shift = torch.empty(args.shift_va_mb << 20, dtype=torch.uint8, device=dev) if args.shift_va_mb else None   # noqa: F841
    torch.manual_seed(1234)
    net = NeRFNetwork(bound=2).to(dev)
    net.encoder.embeddings.data.uniform_(-args.table_scale, args.table_scale)
    r = NeRFRenderer(net, bound=2, min_near=0.2).to(dev)
    r.density_bitfield = torch.from_numpy(S.pack_bits_np(S.flower_density_grid(), 10.0)).to(dev)
    net.eval(); r.eval()
    H, W = 192, 256
    o, d = S.frame_rays(H, W, focal=1111.1 * H / 800, radius=1.6)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)

    def render():
        with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            res = r.render_eval(o, d, bg_color=1, max_steps=1024, max_n_step=1, frame_loop=not args.operator_loop)
        return res["image"].clone()
    first = render()
    torch.cuda.synchronize()
    if args.as_neighbour:
        print("ready", flush=True)
    bad_frames, max_rays, first_runs, max_abs = 0, 0, None, 0.0
    t0 = time.perf_counter()
    for k in range(args.frames):
        img = render()
        ne = (img.view(torch.int32) != first.view(torch.int32)).any(dim=1)
        n = int(ne.sum())
        if n:
            bad_frames += 1
            max_rays = max(max_rays, n)
            max_abs = max(max_abs, float((img - first).abs().max()))
            if first_runs is None:
                first_runs = runs_of(ne.nonzero().flatten().tolist())[:16]
    out = {"lib": os.path.basename(args.lib) or "shipped", "neighbour": "(is the neighbour)" if args.as_neighbour else args.neighbour,
           "table_scale": args.table_scale, "shift_va_mb": args.shift_va_mb,
           "loop": "operator" if args.operator_loop else "frame", "frames": args.frames, "frames_that_differ": bad_frames,
           "max_rays_differing": max_rays, "max_abs_diff": max_abs, "runs_start_length": first_runs, "seconds": round(time.perf_counter() - t0, 1)}
    neighbour_line = None
    if child is not None:
        if args.neighbour == "same":
            try:
                rest, _ = child.communicate(timeout=600)
                neighbour_line = [ln for ln in rest.splitlines() if ln.startswith("{")][-1:]
            except subprocess.TimeoutExpired:
                child.kill()
        else:
            child.kill(); child.wait()
    if neighbour_line:
        out["neighbour_result"] = json.loads(neighbour_line[0])
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
