"""A/B of the sparse-preset train step (bench.sparse_step) -- run with different environment switches:
   LAE_GRID_BWD_BK_TARGET=8 python tools/sparse_ab.py"""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
r = bench.sparse_step(dev)
print(json.dumps({"bk": os.environ.get("LAE_GRID_BWD_BK_TARGET"), "ms_per_step": r["ms_per_step"], "windows": r["windows_ms"], "one_graph": r["one_graph_per_step_ms"],
                  "ops": r["operator_ms_per_step"]}))
