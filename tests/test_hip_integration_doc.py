"""The ctypes binding printed in INTEGRATION.md §2 — what a maintainer of the reference would paste into src/model.py —
is executed as written and must reproduce the package's own path."""
import os
import re
import types

import pytest
import torch

from conftest import REPO
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu


def test_integration_stub_runs_and_matches(monkeypatch):
    assert torch.cuda.is_available()
    from trajectory_optimization_amd import _lib, ops
    _lib.lib()  # make sure the library is built
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = re.search(r"## 2\..*?```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libtrajopt_hip.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    L, Cam, Rewards = ns["_L"], ns["_Cam"], ns["_Rewards"]

    dev = torch.device("cuda:0")
    n, W = 30_000, 7
    pts = torch.from_numpy(synth.make_cloud(n, seed=8)).to(dev)
    poses_np, quats_np = synth.make_path(W, optical=True, jitter_seed=8)
    # "in ModelTraj.__init__, once per model"
    m = types.SimpleNamespace(n=n, npad=L.tohip_padded_points(n))
    m.packed = torch.empty(L.tohip_packed_cloud_bytes(n), dtype=torch.uint8, device=dev)
    scratch = torch.empty(L.tohip_pack_workspace_bytes(n), dtype=torch.uint8, device=dev)
    assert L.tohip_pack_cloud(pts.data_ptr(), n, 1, m.packed.data_ptr(), scratch.data_ptr(), scratch.numel(), ns["_stream"]()) == 0
    m.ws = torch.zeros(L.tohip_traj_workspace_bytes(n, W), dtype=torch.uint8, device=dev)
    import ctypes
    m.cam = Cam((ctypes.c_float * 9)(*synth.K_INTRINS.flatten().tolist()), synth.IMG_WIDTH, synth.IMG_HEIGHT, 1.0, 5.0, 1e-6)
    poses = torch.from_numpy(poses_np).to(dev).requires_grad_(True)
    quats = torch.from_numpy(quats_np).to(dev).requires_grad_(True)
    rewards = Rewards.apply(poses, quats, m)
    w = torch.linspace(0.2, 1.0, n, device=dev)
    (w * rewards).sum().backward()

    cloud = ops.PackedCloud(pts)
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    ws = ops.TrajWorkspace(cloud, W)
    p, q = poses.detach(), quats.detach()
    lo_sum, minmax = ops.traj_forward(cloud, p, q, cam, ws)
    ref_rewards, _ = ops.traj_reward(cloud, lo_sum, cam, ws)
    pg, qg = ops.traj_backward(cloud, W, cam, ws, lo_sum, grad_rewards=w.contiguous())
    assert torch.equal(rewards.detach(), ref_rewards)
    assert torch.equal(poses.grad, pg) and torch.equal(quats.grad, qg)
    assert float(pg.abs().max()) > 0
