#!/usr/bin/env python3
"""Child process of tests/test_gpu_frame.py::test_frame_loop_degrades_*: renders one seeded 200x200 frame with the device-resident
loop under whatever environment the parent set (GPU_MAX_HW_QUEUES=1, LAE_FRAME_OVERLAP=1 + short wait time-out, many live torch
streams) and prints {"sha": sha256 of image/depth/weights_sum, "finite": ..., "seconds": ...} as one JSON line; warnings of
the library go to stderr.  Started fresh: the environment must be in place before the first HIP call."""
import hashlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import bench
    from laenerf_amd import synthetic as S
    dev = torch.device("cuda", 0)
    n_streams = int(os.environ.get("LAE_TEST_LIVE_STREAMS", "0"))
    live = [torch.cuda.Stream() for _ in range(n_streams)]
    for st in live:                                          # make them real: one tiny kernel each
        with torch.cuda.stream(st):
            torch.zeros(8, device=dev).add_(1)
    net, r = bench.eval_model(dev)
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    o, d = S.frame_rays(200, 200)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    run_on = live[len(live) // 2] if live else torch.cuda.current_stream()
    out = []
    for k in range(int(os.environ.get("LAE_TEST_FRAMES", "2"))):
        t0 = time.perf_counter()
        with torch.cuda.stream(run_on), torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
            res = r.render_eval(o, d, bg_color=1, max_steps=1024)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for key in ("image", "depth", "weights_sum"):
            h.update(res[key].float().cpu().numpy().tobytes())
        from laenerf_amd.backend import raymarching_backend
        out.append({"sha": h.hexdigest()[:16], "finite": bool(torch.isfinite(res["image"]).all().item()),
                    "seconds": round(time.perf_counter() - t0, 3), "status": raymarching_backend.render_frame_last_status()})
    print(json.dumps({"frames": out, "mode": raymarching_backend.render_frame_mode()}), flush=True)


if __name__ == "__main__":
    main()
