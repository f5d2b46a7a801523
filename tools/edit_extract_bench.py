"""bench.py's `edit_extract` object alone (configs[4] extraction: extract_views over 8 bonsai-shaped 1080p poses with a grow grid)"""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
print(json.dumps(bench.edit_extract(torch.device("cuda", 0))))
