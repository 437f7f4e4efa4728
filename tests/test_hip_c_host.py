"""The C ABI used from a plain C++ host (examples/c_host.cpp: hipMalloc, no Python, no torch) gives what the Python
path gives on the same inputs — the boundary carries no hidden dependency on torch."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import REPO
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu


def _lcg_inputs(n, W):
    s = np.uint64(12345)
    vals = np.empty(3 * n, np.float32)
    for i in range(3 * n):
        s = (s * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        vals[i] = np.float32(int(s) >> 8) * np.float32(1.0 / 16777216.0)
    pts = vals.reshape(n, 3) * np.array([30, 30, 4], np.float32) - np.array([15, 15, 2], np.float32)
    poses = np.zeros((W, 3), np.float32)
    for w in range(W):
        poses[w, 0] = np.float32(-8.0) + np.float32(16.0) * np.float32(w) / np.float32(max(W - 1, 1))
        poses[w, 1] = np.float32(0.5) * np.float32(w)
    quats = np.tile(np.array([[0.5, -0.5, 0.5, -0.5]], np.float32), (W, 1))
    return pts, poses, quats


def test_cpp_host_matches_python_path(tmp_path):
    assert torch.cuda.is_available()
    exe = tmp_path / "c_host"
    libdir = os.path.join(REPO, "trajectory_optimization_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-I" + os.path.join(REPO, "include"),
                           os.path.join(REPO, "examples", "c_host.cpp"), "-L" + libdir, "-ltrajopt_hip",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    n, W = 6000, 5
    out = json.loads(subprocess.check_output([str(exe), str(n), str(W)], text=True).strip().splitlines()[-1])
    from trajectory_optimization_amd import ops
    dev = torch.device("cuda:0")
    pts, poses, quats = _lcg_inputs(n, W)
    cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    ws = ops.TrajWorkspace(cloud, W)
    lo_sum, minmax = ops.traj_forward(cloud, p, q, cam, ws)
    rewards, scalars = ops.traj_reward(cloud, lo_sum, cam, ws)
    pg, qg = ops.traj_backward(cloud, W, cam, ws, lo_sum, scalars=scalars, gout=torch.ones(1, device=dev))
    # same library, same inputs, same launch sequence: identical bits
    assert np.float32(out["mean_reward"]) == scalars[0].item() and np.float32(out["loss_vis"]) == scalars[1].item()
    assert np.array_equal(np.asarray(out["poses_grad"], np.float32).reshape(W, 3), pg.cpu().numpy())
    assert np.array_equal(np.asarray(out["quats_grad"], np.float32).reshape(W, 4), qg.cpu().numpy())
    assert 0.5 < out["mean_reward"] < 1.0 and np.abs(pg.cpu().numpy()).max() > 0
    # the fused call of the same host: rewards and scalars bitwise the separate calls', gradients those of the Python fused path
    f = out["fused"]
    assert f["rewards_equal"] is True and np.float32(f["mean_reward"]) == scalars[0].item() and np.float32(f["loss_vis"]) == scalars[1].item()
    r2 = ops.traj_forward_backward(cloud, p, q, cam, ws, torch.ones(1, device=dev))
    assert np.array_equal(np.asarray(f["poses_grad"], np.float32).reshape(W, 3), r2[2].cpu().numpy())
    assert np.array_equal(np.asarray(f["quats_grad"], np.float32).reshape(W, 4), r2[3].cpu().numpy())
