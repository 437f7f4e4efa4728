#!/usr/bin/env python3
"""Host time of each phase of the drop-in loop (no synchronisation inside the loop: what the issuing thread spends).  GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import synth
from trajectory_optimization_amd.model import ModelTraj
from trajectory_optimization_amd.optimizer import Adam as HipAdam

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
pts = synth.make_cloud(n, seed=0)
poses, quats = synth.make_path(128, optical=True)
m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS), synth.IMG_WIDTH,
              synth.IMG_HEIGHT, device=dev)
for oname, opt in (("torch.optim.Adam", torch.optim.Adam([{"params": [m.poses], "lr": 0.1}, {"params": [m.quats], "lr": 0.02}])),
                   ("HipAdam", HipAdam([{"params": [m.poses], "lr": 0.1}, {"params": [m.quats], "lr": 0.02}]))):
    t = np.zeros(4)
    N = 300
    for i in range(N + 20):
        if i % 50 == 0:
            torch.cuda.synchronize()
        a = time.perf_counter(); opt.zero_grad()
        b = time.perf_counter(); loss = m(vis_wps_dist=0.0)
        c = time.perf_counter(); loss.backward()
        d = time.perf_counter(); opt.step()
        e = time.perf_counter()
        if i >= 20:
            t += (b - a, c - b, d - c, e - d)
    torch.cuda.synchronize()
    print(oname, "host us: zero_grad %.1f model() %.1f backward() %.1f step() %.1f total %.1f" % (*(1e6 * t / N), 1e6 * t.sum() / N))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    opt.zero_grad(); loss = m(vis_wps_dist=0.0); loss.backward(); opt.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
