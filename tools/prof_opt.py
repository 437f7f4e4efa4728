#!/usr/bin/env python3
"""The device-resident optimisation loop (optimizer.optimize_trajectory: one library call and five launches per step) on the
BASELINE workload, for `rocprofv3 --kernel-trace --stats` (tools/collect_profiles.sh -> profiles/r04_optimize_kernel_stats.csv):

    rocprofv3 --kernel-trace --stats -d gpurun_out/po -o o -- python3 tools/prof_opt.py --steps 120

Prints the wall time per step of the whole run and of its last third (the step gets slower as the trajectory moves: more
flagged pairs), and what the last step's forward found."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import ops, synth  # noqa: E402
from trajectory_optimization_amd.model import ModelTraj  # noqa: E402
from trajectory_optimization_amd.optimizer import _OptRun  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--wps", type=int, default=128)
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--lr-pose", type=float, default=0.1)
ap.add_argument("--lr-quat", type=float, default=0.02)
args = ap.parse_args()
dev = torch.device("cuda:0")
pts = torch.from_numpy(synth.make_cloud(args.points, seed=0)).to(dev)
poses, quats = synth.make_path(args.wps, optical=True)
m = ModelTraj(pts, torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS), synth.IMG_WIDTH, synth.IMG_HEIGHT,
              device=dev, dense=args.dense)
run = _OptRun([m], args.steps, args.lr_pose, args.lr_quat, 1e9, 1e9, 0.0, (0.9, 0.999), 1e-8)
idx = dev.index or 0
for i in range(3):   # (the first calls pay module load and clock ramp; they are part of the trace, not of the wall times below)
    run.fn(run.ref, i, torch._C._cuda_getCurrentRawStream(idx))
torch.cuda.synchronize(dev)
third = args.steps // 3
t0 = time.perf_counter()
idx = dev.index or 0
for i in range(3, args.steps):
    if i == args.steps - third:
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
    run.fn(run.ref, i, torch._C._cuda_getCurrentRawStream(idx))
torch.cuda.synchronize(dev)
t2 = time.perf_counter()
st = ops.traj_step_stats(m._cloud, run.ws)
res = run.results(args.steps)[0]
print(f"steps {args.steps}: {1e3 * (t2 - t0) / (args.steps - 3):.4f} ms/step overall, {1e3 * (t2 - t1) / third:.4f} ms/step over the last {third}; "
      f"loss {res.losses[0]:.4f} -> {res.losses[-1]:.4f}; last step: {st['flagged_pairs']} flagged pairs "
      f"({100 * st['flagged_fraction']:.2f} %), {st['candidate_slots']} candidate slots, {st['evaluated_pairs']} pairs evaluated")
