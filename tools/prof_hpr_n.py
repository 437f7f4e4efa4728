import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_optimization_amd import synth, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")
P = torch.from_numpy(synth.make_cloud(n, seed=0)).to(dev)
for _ in range(2):
    ops.hidden_pts_removal(P)
torch.cuda.synchronize()
