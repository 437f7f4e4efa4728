"""The occlusion rows' refresh alone (1 M points x 128 waypoints), a few times per method: for rocprofv3 --kernel-trace --stats.
    python tools/prof_occlusion.py [hpr|zbuffer] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import synth
from trajectory_optimization_amd.model import ModelTraj

method = sys.argv[1] if len(sys.argv) > 1 else "zbuffer"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
pts = torch.from_numpy(synth.make_cloud(1_000_000, seed=0))
poses, quats = synth.make_path(128, optical=True)
m = ModelTraj(pts, torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS), synth.IMG_WIDTH, synth.IMG_HEIGHT,
              device=dev, occlusion=method, occlusion_refresh_every=1000)
with torch.no_grad():
    m()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m.refresh_occlusion()
    with torch.no_grad():
        m()
torch.cuda.synchronize()
print(f"{method}: refresh + one forward {1e3 * (time.perf_counter() - t0) / reps:.2f} ms")
