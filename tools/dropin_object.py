import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
print(json.dumps(bench.drop_in_step(dev, 0.3556)))
