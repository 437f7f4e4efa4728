#!/usr/bin/env python3
"""Accuracy of the occlusion refresh policy (ModelTraj(occlusion=..., occlusion_refresh_every=k)) on the bundled
cloud: 30 optimiser steps with the masks rebuilt every k-th forward against the same run with k = 1; both final trajectories are
evaluated with FRESH masks.  GPU box:   python3 tools/occlusion_refresh_accuracy.py   -> profiles/r04_occlusion_refresh_accuracy.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import synth
from trajectory_optimization_amd.model import ModelTraj
from trajectory_optimization_amd.optimizer import optimize_trajectory
dev = torch.device("cuda:0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = np.load(os.path.join(REPO, "tests", "golden", "bundled.npz"))
pts, poses = torch.from_numpy(b["pts"].astype(np.float32)), torch.from_numpy(b["poses"].astype(np.float32))
quats = torch.tensor([[1.0, 0, 0, 0]]).repeat(len(poses), 1)
for method in ("hpr", "zbuffer"):
    out = {}
    for k in (1, 5, 10, 30):
        m = ModelTraj(pts, poses, quats, torch.from_numpy(synth.K_INTRINS), synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev, occlusion=method, occlusion_refresh_every=k)
        r = optimize_trajectory(m, n_opt_steps=30, lr_pose=0.12, lr_quat=0.05, rewards_th=1e9)
        # evaluate the final trajectory with FRESH masks
        m.refresh_occlusion(); m.occlusion_refresh_every = 1
        with torch.no_grad():
            loss = m()
        out[k] = (float(loss), float(m.rewards.mean()), m.poses.data.clone())
    ref = out[1]
    for k in (5, 10, 30):
        print(method, "refresh_every", k, "final loss (fresh masks) %.5f vs %.5f (k=1): rel %.2e; mean reward %.6f vs %.6f; max waypoint distance to the k=1 run %.4f m" % (
            out[k][0], ref[0], abs(out[k][0] - ref[0]) / ref[0], out[k][1], ref[1], float((out[k][2] - ref[2]).norm(dim=1).max())))
