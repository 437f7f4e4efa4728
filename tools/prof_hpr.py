import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_optimization_amd import synth, ops
dev = torch.device("cuda:0")
P = torch.from_numpy(synth.make_cloud(1_000_000, seed=0)).to(dev)
for _ in range(2):
    ops.hidden_pts_removal(P)
torch.cuda.synchronize()
