#!/usr/bin/env python3
"""Where the small kernels of a step spend their time (GPU box).

Builds the library once more with -DTOHIP_STAMPS (thread 0 of every block notes the 100 MHz real-time counter at a few places of
k_traj_probe / _pass1_cull / _sparse / _pairs / _finish; the shipped library has none of this), runs the 1 M x 128 step and prints,
per kernel: the span from its first block's start to its last block's end, how far apart the blocks start, and the timeline of the
median and of the slowest block (ns from the block's own start).

    python tools/kernel_timeline.py [culled|dense] [extent_xy] [moved_steps] [waypoints] [trajectories] > profiles/rNN_small_kernel_timelines.txt

waypoints (default 128) per trajectory; trajectories > 1: that many trajectories side by side (0.8 m apart, tools/time_multi.py's
set) as ONE batch of virtual waypoints (tohip_traj_forward_backward_multi) — the large-W regime: 1 024 virtual waypoints either
as one trajectory (BASELINE config 4's work on one GPU) or as eight.

moved_steps > 0: the trajectory first takes that many optimiser steps (optimize_trajectory, lr 0.1 / 0.02): the step a run ends
with, not the one it starts with (more flagged pairs in more candidate slots).
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trajectory_optimization_amd import _lib  # noqa: E402

KERNELS = ["k_traj_probe", "k_traj_pass1_cull", "k_traj_sparse", "k_traj_pairs", "k_traj_finish"]
LABELS = {
    "k_traj_probe": ["start", "record built, samples requested", "sample evaluated and reduced", "bounds, cull distance, resets",
                     "reachable slots (culled mode)"],
    "k_traj_pass1_cull": ["start", "row of reachable slots in LDS", "prefix and list", "wave 0's pairs evaluated", "every wave's",
                          "end"],
    "k_traj_sparse": ["start", "candidate known", "flags, staged records", "wave 0's share of the forward sweep", "every wave's",
                      "sums, rewards", "pair list's answer", "end"],
    "k_traj_pairs": ["start", "list length, first pair", "end", "the block's last wave"],
    "k_traj_finish": ["start", "flag words, tie slots", "wave 0: tie slots re-evaluated", "every wave: ties, rows, group sums",
                      "64 groups added, factor", "end"],
}
BLOCKS, N = 1024, 12


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "culled"
    xy = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
    moved = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    wps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    n_traj = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    so = os.path.join(tempfile.gettempdir(), "libtrajopt_stamps.so")
    subprocess.check_call([_lib.HIPCC] + _lib.HIPCC_FLAGS + ["-DTOHIP_STAMPS", _lib.SRC, "-o", so])
    _lib.LIB_PATH = so   # before the first lib() call: this process runs the diagnostic build
    from trajectory_optimization_amd import ops, synth
    import bench
    L = _lib.lib()
    L.tohip_stamps_read.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    pts = synth.make_cloud(bench.N_POINTS, seed=0, extent=(xy, xy, 4.0))
    wps = wps or bench.WPS_PER_GPU
    poses, quats = synth.make_path(wps, optical=True, scale=xy / 40.0)
    if n_traj > 1:
        poses = np.concatenate([poses + np.float32([0.0, 0.8 * i - 0.4 * n_traj, 0.0]) for i in range(n_traj)])
        quats = np.concatenate([quats] * n_traj)
    cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    if moved > 0:
        from trajectory_optimization_amd.model import ModelTraj
        from trajectory_optimization_amd.optimizer import optimize_trajectory
        m = ModelTraj(torch.from_numpy(pts).to(dev), p, q, torch.from_numpy(synth.K_INTRINS), synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev)
        optimize_trajectory(m, n_opt_steps=moved, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=0.0)
        p, q = m.poses.data.clone(), m.quats.data.clone()
    ws = ops.TrajWorkspace(cloud, wps * n_traj, n_traj)
    gout = torch.ones(n_traj, device=dev)
    toff = (torch.arange(n_traj + 1, dtype=torch.int32) * wps).to(dev)
    flags = ops.DENSE if mode == "dense" else 0
    buf = (ctypes.c_ulonglong * (len(KERNELS) * BLOCKS * N))()
    runs = []
    for it in range(16):
        L.tohip_stamps_clear()
        if n_traj > 1:
            ops.traj_forward_backward_multi(cloud, p, q, toff, cam, ws, gout, flags=flags)
        else:
            ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=flags)
        torch.cuda.synchronize()
        L.tohip_stamps_read(buf)
        if it >= 6:
            runs.append(np.array(buf[:], dtype=np.int64).reshape(len(KERNELS), BLOCKS, N))
    st = ops.traj_step_stats(cloud, ws)
    print(f"# {bench.N_POINTS} points x {n_traj} x {wps} waypoints, {mode}, {xy:g} x {xy:g} x 4 m" + (f", after {moved} optimiser steps" if moved else "") +
          f" ({st['flagged_pairs']} flagged pairs in {st['candidate_slots']} candidate slots); ns, 100 MHz counter (10 ns steps); "
          f"thread 0 of each block; last of {len(runs)} stamped steps")
    a = runs[-1]
    for ki, name in enumerate(KERNELS):
        s = a[ki]
        ran = s[:, 0] > 0
        if not ran.any():
            continue
        s = s[ran]
        labels = LABELS[name]
        t0 = s[:, 0].min()
        ends = s.max(axis=1)   # the last place a block stamped
        print(f"{name}: {ran.sum()} blocks stamped; first start -> last end {(ends.max() - t0) * 10} ns; starts spread over "
              f"{(s[:, 0].max() - t0) * 10} ns")
        dur = (ends - s[:, 0]) * 10
        order = np.argsort(dur)
        rel_end = np.sort((ends - t0) * 10)
        print("    block ends after the kernel's first start, ns: " + ", ".join(f"{int(q * 100)} % {int(rel_end[min(len(rel_end) - 1, int(q * len(rel_end)))])}"
                                                                                   for q in (0.25, 0.5, 0.75, 0.9, 0.99)) +
              "; block lifetimes, ns: " + ", ".join(f"{int(q * 100)} % {int(dur[order[min(len(order) - 1, int(q * len(order)))]])}" for q in (0.25, 0.5, 0.75, 0.9, 0.99)))
        for what, b in (("median block", order[len(order) // 2]), ("slowest block", order[-1])):
            r = s[b]
            line = ", ".join(f"{labels[i]} {int((r[i] - r[0]) * 10)}" for i in range(1, len(labels)) if r[i] > 0)
            print(f"    {what} (starts at +{(r[0] - t0) * 10} ns): {line}")


if __name__ == "__main__":
    main()
