"""the frame objects of bench.py's default line alone (eval_frame, frame1080 incl. shard_of_8): python tools/bench_frames.py"""
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
e = bench.eval_frame(dev)
f = bench.frame1080(dev)
print(json.dumps({"eval_frame": {k: e[k] for k in ("ms_per_frame", "operator_loop_ms", "iterations", "max_abs_image_diff_vs_operator_loop")},
                  "frame1080": {"ms_per_frame": f["ms_per_frame"], "shard_of_8_ms": f["shard_of_8"]["ms"], "shard_ref_ms": f["shard_of_8"]["reference_schedule_ms"]}}))
