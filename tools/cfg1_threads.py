"""cpu_baseline_cfg1 (bench.py) at 16 / 32 / 64 / 128 / 256 host threads: how the cfg1 CPU baseline scales on a box (round 4)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import bench, torch
for n in (16, 32, 64, 128, 256):
    t = time.time()
    r = bench.cpu_baseline_cfg1(n, budget_s=2.0)
    print(n, r["ms_per_step"], r["steps"], round(time.time() - t, 1), flush=True)
