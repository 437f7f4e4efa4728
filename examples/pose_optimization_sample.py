#!/usr/bin/env python3
"""The reference's stand-alone camera pose optimisation sample (/root/reference/src/pose_optimization_sample.py)
without ROS: load a `.npz` cloud, optimise one camera pose (start (6, 2, 0), random orientation) for the soft
visibility of the cloud with Adam + ExponentialLR(0.95) stepped every N/10 iterations, write pose and per-point
observations to an .npz (what the reference publishes as odometry / tf / an intensity cloud).

    python examples/pose_optimization_sample.py --points point_cloud_10.npz [--hpr]
    python examples/pose_optimization_sample.py            # the bundled sample kept as a test fixture
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

from trajectory_optimization_amd.model import ModelPose  # noqa: E402
from trajectory_optimization_amd.samples import load_data  # noqa: E402
from trajectory_optimization_amd.tools import load_intrinsics  # noqa: E402


def random_quaternion(seed):
    """pytorch3d.transforms.random_quaternions(1): a normalised Gaussian 4-vector with a non-negative real part (wxyz)."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn((1, 4), generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    return q * torch.where(q[:, :1] < 0, -1.0, 1.0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", default=None, help="point_cloud_<i>.npz (key 'pts')")
    ap.add_argument("--opt-steps", type=int, default=400)
    ap.add_argument("--lr-pose", type=float, default=0.1)
    ap.add_argument("--lr-quat", type=float, default=0.1)
    ap.add_argument("--hpr", action="store_true", help="multiply the observations by the world-frame HPR mask (model.py:114)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="pose_opt_result.npz")
    args = ap.parse_args(argv)

    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device: the visibility path has no CPU fallback")
    device = torch.device("cuda:0")
    if args.points is None:
        pts_np = np.load(os.path.join(REPO, "tests", "golden", "bundled.npz"))["pts"]
    else:
        pts_np, _, _ = load_data(args.points)
    K, img_width, img_height = load_intrinsics(device=device)
    model = ModelPose(points=torch.from_numpy(pts_np), trans0=torch.tensor([[6.0, 2.0, 0.0]]), q0=random_quaternion(args.seed),
                      intrins=K, img_width=img_width, img_height=img_height, min_dist=1.0, max_dist=5.0, device=device)
    optimizer = torch.optim.Adam([{"params": [model.trans], "lr": args.lr_pose},
                                  {"params": [model.quat], "lr": args.lr_quat}])
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=0.95)
    every = max(1, args.opt_steps // 10)
    losses, t_step = [], 0.0
    for i in range(args.opt_steps):
        t0 = time()
        optimizer.zero_grad()
        loss = model(hpr=args.hpr)
        loss.backward()
        optimizer.step()
        if i % every == 0:
            scheduler.step()
        losses.append(loss.item())
        t_step += time() - t0
    quat = F.normalize(model.quat.detach())
    np.savez_compressed(args.out, trans=model.trans.detach().cpu().numpy(), quat_wxyz=quat.cpu().numpy(),
                        observations=model.observations.detach().cpu().numpy(), losses=np.asarray(losses, dtype=np.float32))
    print(f"{args.opt_steps} steps, {1e3 * t_step / max(args.opt_steps, 1):.2f} ms/step; loss {losses[0]:.3e} -> {losses[-1]:.3e} "
          f"(sum of observations {1 / losses[0]:.1f} -> {1 / losses[-1]:.1f}); wrote {args.out}")
    return losses


if __name__ == "__main__":
    main()
