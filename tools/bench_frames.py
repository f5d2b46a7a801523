"""the frame objects of bench.py's default line alone (eval_frame, frame1080 incl. shards_of_8 + exchange): python tools/bench_frames.py"""
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
e = bench.eval_frame(dev)
f = bench.frame1080(dev)
print(json.dumps({"eval_frame": {k: e[k] for k in ("ms_per_frame", "operator_loop_ms", "iterations", "max_abs_image_diff_vs_operator_loop")},
                  "frame1080": {k: f[k] for k in ("ms_per_frame", "best_n1_ms_per_frame", "row_budget_ms", "shards_of_8", "exchange", "projected_speedup_at_8")}}))
