#!/usr/bin/env python3
"""Step time of the drop-in path (ModelTraj + torch.optim.Adam, as the reference's loops use it) next to the launch-only
optimizer.optimize_trajectory, on the reference's bundled sample and on the bench workload.  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import synth
from trajectory_optimization_amd.model import ModelTraj
from trajectory_optimization_amd.optimizer import optimize_trajectory, Adam
from trajectory_optimization_amd.tools import load_intrinsics

dev = torch.device("cuda:0")
K, iw, ih = load_intrinsics(dev)
b = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bundled.npz"))
cases = {"bundled 40k x 27 wps": (b["pts"], b["poses"], np.tile(np.array([[1, 0, 0, 0]], np.float32), (len(b["poses"]), 1)), 0.5),
         "synthetic 1M x 128 wps": (synth.make_cloud(1_000_000, seed=0),) + synth.make_path(128, optical=True) + (0.0,)}
for name, (pts, poses, quats, vwd) in cases.items():
    def model():
        return ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), K, iw, ih, device=dev)
    m = model()
    opt = torch.optim.Adam([{"params": [m.poses], "lr": 0.1}, {"params": [m.quats], "lr": 0.02}])
    tf = tb = to = 0.0
    n = 60
    for i in range(n + 5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(); loss = m(vis_wps_dist=vwd)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if i >= 5:
            tf += t1 - t0; tb += t2 - t1; to += t3 - t2
    m3 = model()
    opt3 = Adam([{"params": [m3.poses], "lr": 0.1}, {"params": [m3.quats], "lr": 0.02}])
    for i in range(5):
        opt3.zero_grad(); m3(vis_wps_dist=vwd).backward(); opt3.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        opt3.zero_grad(); m3(vis_wps_dist=vwd).backward(); opt3.step()
    torch.cuda.synchronize(); t_fused_adam = (time.perf_counter() - t0) / n
    m2 = model()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    optimize_trajectory(m2, n_opt_steps=n, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=vwd)
    torch.cuda.synchronize(); tl = time.perf_counter() - t0
    print(f"{name}: drop-in forward {1e3*tf/n:.3f} + backward {1e3*tb/n:.3f} + Adam {1e3*to/n:.3f} = {1e3*(tf+tb+to)/n:.3f} ms/step; "
          f"same loop with optimizer.Adam, no per-phase syncs {1e3*t_fused_adam:.3f} ms/step; launch-only loop {1e3*tl/n:.3f} ms/step")
