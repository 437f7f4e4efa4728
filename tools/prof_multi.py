#!/usr/bin/env python3
"""The large-W regime on ONE GPU (the N = 1 points of BASELINE configs 2, 4 and 5, and eight concurrent trajectories), for
`rocprofv3 --kernel-trace --stats` (tools/collect_profiles.sh -> profiles/r05_multi_*_kernel_stats.csv) and for wall clocks:

    rocprofv3 --kernel-trace --stats -d gpurun_out/pm -o m -- python3 tools/prof_multi.py --scenario multi8 --mode culled

scenarios
  multi8   optimizer.optimize_trajectories: 8 trajectories x 128 waypoints over one 1 M-point cloud (tohip_traj_opt_step, n_traj = 8)
  w1024    config 4's work on one GPU: 1 M points x 1 024 waypoints, tohip_traj_forward_backward
  cam5     config 5's work on one GPU: 5 cameras x 1 M points x 256 waypoints = 1 280 virtual waypoints
  c2       config 2: 100 k points x 32 waypoints, forward + reward only (tohip_traj_forward + tohip_traj_reward)
  c3       config 3 (the headline): 1 M x 128, tohip_traj_forward_backward
Prints one JSON object per (scenario, mode): ms/step (wall, K steps between synchronisations), evaluations/s, per-kernel-class
microseconds (HIP events on the launch stream, a separate pass), what the forward found."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import _lib, ops, synth  # noqa: E402


def kernel_us(L, fn, steps):
    ms, cnt = (ctypes.c_double * 6)(), (ctypes.c_int64 * 6)()
    torch.cuda.synchronize()
    L.tohip_profile_enable(1)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    _lib.check(L.tohip_profile_read(ms, cnt), "tohip_profile_read")
    L.tohip_profile_enable(0)
    return {L.tohip_profile_name(i).decode(): 1e3 * ms[i] / cnt[i] for i in range(6) if cnt[i] > 0}


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def scenario(name, mode, dev, steps, warmup, events=True):
    L = _lib.lib()
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    flags = ops.DENSE if mode == "dense" else 0
    gout = torch.ones(1, device=dev)
    if name == "multi8":
        from trajectory_optimization_amd.model import ModelTraj
        from trajectory_optimization_amd.optimizer import _OptRun
        n, w, B = 1_000_000, 128, 8
        pts = torch.from_numpy(synth.make_cloud(n, seed=0)).to(dev)
        K = torch.from_numpy(synth.K_INTRINS)
        models = []
        for i in range(B):
            p, q = synth.make_path(w, optical=True)
            models.append(ModelTraj(pts, torch.from_numpy(p + np.float32([0.0, 0.8 * i - 0.4 * B, 0.0])), torch.from_numpy(q), K, synth.IMG_WIDTH,
                                    synth.IMG_HEIGHT, device=dev, dense=mode == "dense", cloud=models[0] if models else None))
        total = warmup + 2 * steps + 8
        run = _OptRun(models, total, 0.02, 0.005, 1e9, 1e9, 0.0, (0.9, 0.999), 1e-8)
        it = iter(range(total))
        idx = dev.index or 0

        def fn():
            _lib.check(run.fn(run.ref, next(it), torch._C._cuda_getCurrentRawStream(idx)), "tohip_traj_opt_step")
        evals = n * w * B
        cloud, ws = models[0]._cloud, run.ws
        what = f"{B} trajectories x {w} waypoints x {n} points: tohip_traj_opt_step (n_traj = {B}), one call and five launches per step"
    elif name in ("w1024", "cam5", "c3"):
        n = 1_000_000
        w, cams = {"w1024": (1024, 1), "cam5": (256, 5), "c3": (128, 1)}[name]
        cloud = ops.PackedCloud(torch.from_numpy(synth.make_cloud(n, seed=0)).to(dev))
        p, q = synth.make_path(w, optical=True)
        p, q = torch.from_numpy(p).to(dev), torch.from_numpy(q).to(dev)
        rig = ops.CameraRig(*synth.camera_rig(cams), dev) if cams > 1 else None
        ws = ops.TrajWorkspace(cloud, w * cams)
        lo, mm, rw = torch.empty(cloud.npad, device=dev), torch.empty((w * cams, 2), device=dev), torch.empty(n, device=dev)

        def fn():
            ops.traj_forward_backward(cloud, p, q, cam, ws, gout, rig=rig, flags=flags, lo_sum=lo, minmax=mm, rewards=rw)
        evals = n * w * cams
        what = f"{n} points x {w} waypoints" + (f" x {cams} cameras" if cams > 1 else "") + ": tohip_traj_forward_backward, five launches"
    elif name == "c2":
        n, w = 100_000, 32
        cloud = ops.PackedCloud(torch.from_numpy(synth.make_cloud(n, seed=0)).to(dev))
        p, q = synth.make_path(w, optical=True)
        p, q = torch.from_numpy(p).to(dev), torch.from_numpy(q).to(dev)
        ws = ops.TrajWorkspace(cloud, w)
        lo, mm, rw, sc = torch.empty(cloud.npad, device=dev), torch.empty((w, 2), device=dev), torch.empty(n, device=dev), torch.empty(4, device=dev)

        def fn():
            ops.traj_forward(cloud, p, q, cam, ws, flags=flags, lo_sum=lo, minmax=mm, rewards_half=rw)
            ops.traj_reward(cloud, lo, cam, ws, rewards=rw, scalars=sc, prefilled=True)
        evals = n * w
        what = f"{n} points x {w} waypoints, forward + reward only: tohip_traj_forward + tohip_traj_reward, four launches"
    else:
        raise SystemExit(f"unknown scenario {name}")
    ms = timed(fn, steps, warmup)
    out = {"scenario": name, "mode": mode, "what": what, "ms_per_step": ms, "evals_per_s": evals / (ms * 1e-3), "steps": steps}
    if events:
        out["kernel_us"] = kernel_us(L, fn, min(steps, 8))
    st = ops.traj_step_stats(cloud, ws)
    out.update(flagged_pairs=st["flagged_pairs"], candidate_slots=st["candidate_slots"], slots=st["slots"], virtual_waypoints=st["virtual_waypoints"],
               evaluated_pairs_culled=st["evaluated_pairs"] if mode == "culled" else None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenario", default="multi8,w1024,cam5,c2,c3")
    ap.add_argument("--mode", default="culled,dense")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-events", action="store_true", help="skip the HIP-event pass (under rocprofv3 the trace has the per-kernel times)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for sc in args.scenario.split(","):
        for mode in args.mode.split(","):
            print(json.dumps(scenario(sc, mode, dev, args.steps, args.warmup, events=not args.no_events)), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
