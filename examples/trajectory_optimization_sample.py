#!/usr/bin/env python3
"""The reference's stand-alone trajectory optimisation sample (/root/reference/src/trajectory_optimization_sample.py)
without ROS: load a `.npz` cloud + path, optimise the waypoints for visibility with Adam (two learning rates, an
ExponentialLR decay every N/10 steps), stop on the same gain thresholds, write the result to an .npz.

    python examples/trajectory_optimization_sample.py --points point_cloud_10.npz --poses path_poses_10.npz
    python examples/trajectory_optimization_sample.py            # the bundled sample kept as a test fixture
"""
import argparse
import os
import sys
from time import time

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from trajectory_optimization_amd.model import ModelTraj  # noqa: E402
from trajectory_optimization_amd.samples import load_data, save_result  # noqa: E402
from trajectory_optimization_amd.tools import load_intrinsics  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", default=None, help="point_cloud_<i>.npz (key 'pts')")
    ap.add_argument("--poses", default=None, help="path_poses_<i>.npz (key 'poses')")
    ap.add_argument("--opt-steps", type=int, default=400)
    ap.add_argument("--smooth-weight", type=float, default=14.0)
    ap.add_argument("--length-weight", type=float, default=0.02)
    ap.add_argument("--lr-pose", type=float, default=0.1)
    ap.add_argument("--lr-quat", type=float, default=0.02)
    ap.add_argument("--rewards-th", type=float, default=1.1)
    ap.add_argument("--smoothness-th", type=float, default=0.9)
    ap.add_argument("--out", default="traj_opt_result.npz")
    args = ap.parse_args(argv)

    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device: the visibility path has no CPU fallback")
    device = torch.device("cuda:0")
    if args.points is None:
        d = np.load(os.path.join(REPO, "tests", "golden", "bundled.npz"))
        pts_np, poses_np = d["pts"], d["poses"]
        quats_np = np.tile(np.array([[1.0, 0.0, 0.0, 0.0]], dtype=np.float32), (len(poses_np), 1))
    else:
        pts_np, poses_np, quats_np = load_data(args.points, args.poses)
    K, img_width, img_height = load_intrinsics(device=device)
    model = ModelTraj(points=torch.from_numpy(pts_np), wps_poses=torch.from_numpy(poses_np),
                      wps_quats=torch.from_numpy(quats_np), intrins=K, img_width=img_width, img_height=img_height,
                      smoothness_weight=args.smooth_weight, traj_length_weight=args.length_weight, device=device)
    optimizer = torch.optim.Adam([{"params": [model.poses], "lr": args.lr_pose},
                                  {"params": [model.quats], "lr": args.lr_quat}])
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=0.9)
    log = {"visibility": [], "smoothness": []}
    reward0 = smooth0 = None
    t_step, steps = 0.0, 0
    every = max(1, args.opt_steps // 10)
    for i in range(args.opt_steps):
        t0 = time()
        optimizer.zero_grad()
        loss = model()
        loss.backward()
        optimizer.step()
        if i % every == 0:
            scheduler.step()
        reward, smooth = torch.mean(model.rewards).item(), float(model.loss["smooth"])
        t_step += time() - t0
        steps += 1
        if reward0 is None:
            reward0, smooth0 = reward, smooth
        log["visibility"].append(reward / reward0)
        log["smoothness"].append(smooth0 / smooth)
        if reward / reward0 > args.rewards_th and smooth0 / smooth > args.smoothness_th:
            break  # trajectory_optimization_sample.py: OPTIMIZATION_COMPLETE
    quats = F.normalize(model.quats.detach())
    save_result(args.out, model.poses.detach().cpu().numpy(), quats.cpu().numpy(), model.rewards.detach().cpu().numpy(), log)
    print(f"{steps} steps, {1e3 * t_step / steps:.2f} ms/step; visibility gain {log['visibility'][-1]:.4f}, "
          f"smoothness gain {log['smoothness'][-1]:.4f}; wrote {args.out}")
    return log


if __name__ == "__main__":
    main()
