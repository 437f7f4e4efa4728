"""optimize_trajectories (8 x 128 waypoints, 1 M points, culled) alone: for rocprofv3 --kernel-trace --stats."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import synth
from trajectory_optimization_amd.model import ModelTraj
from trajectory_optimization_amd.optimizer import optimize_trajectories
dev = torch.device("cuda:0")
n, w, B, steps = 1_000_000, 128, int(os.environ.get("B", 8)), 20
pts = torch.from_numpy(synth.make_cloud(n, seed=0))
K = torch.from_numpy(synth.K_INTRINS)
paths = []
for i in range(B):
    p, q = synth.make_path(w, optical=True)
    paths.append((torch.from_numpy(p + np.float32([0.0, 0.8 * i - 0.4 * B, 0.0])), torch.from_numpy(q)))
ms = [ModelTraj(pts, p, q, K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev) for p, q in paths]
optimize_trajectories(ms, n_opt_steps=steps, lr_pose=0.02, lr_quat=0.005, rewards_th=1e9, vis_wps_dist=0.0)
torch.cuda.synchronize()
