import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from trajectory_optimization_amd import synth, ops
dev = torch.device("cuda:0")
n = int(os.environ.get("N", 1_000_000))
pts = synth.make_cloud(n, seed=0)
P = torch.from_numpy(pts).to(dev)
ops.hidden_pts_removal(P)
torch.cuda.synchronize()
os.environ["TOHIP_HULL_TRACE"] = "1"
t = time.perf_counter()
idx, _ = ops.hidden_pts_removal(P)
torch.cuda.synchronize()
print(f"n={n} visible={idx.numel()} ms={(time.perf_counter()-t)*1e3:.2f}")
