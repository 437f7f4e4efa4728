"""Two ranks sharing the one GPU of the test box (gloo rendezvous, host-staged all-reduce): the real sharded
ModelTraj path — HIP kernels per rank, all-reduce of the log-odds vector, gradient assembly — must reproduce
the single-process model.  (RCCL itself needs one GPU per rank; the driver exercises it with bench.py --gpus N.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from trajectory_optimization_amd import synth
    from trajectory_optimization_amd.distributed import WaypointShard, init_from_env
    from trajectory_optimization_amd.model import ModelTraj
    _, _, device = init_from_env(backend="gloo")
    pts = synth.make_cloud(60_000, seed=9)
    poses, quats = synth.make_path(9, optical=True, jitter_seed=9)
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                  synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device, shard=WaypointShard())
    loss = m(vis_wps_dist=0.0)
    loss.backward()
    # the launch-only optimiser on the sharded model: local forward/backward rows, all-reduced log-odds and gradient rows,
    # replicated step remainder
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    m2 = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                   synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device, shard=WaypointShard())
    res = optimize_trajectory(m2, n_opt_steps=4, lr_pose=0.05, lr_quat=0.01, rewards_th=1e9, vis_wps_dist=0.0)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=loss.item(), rewards=m.rewards.detach().cpu().numpy(),
             pg=m.poses.grad.cpu().numpy(), qg=m.quats.grad.cpu().numpy(), opt_poses=m2.poses.detach().cpu().numpy(),
             opt_quats=m2.quats.detach().cpu().numpy(), opt_losses=np.asarray(res.losses))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_model_equals_single_process(tmp_path):
    sys.path.insert(0, REPO)
    from trajectory_optimization_amd import synth
    from trajectory_optimization_amd.model import ModelTraj
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    for k in ("loss", "rewards", "pg", "qg", "opt_poses", "opt_quats", "opt_losses"):
        assert np.array_equal(r0[k], r1[k]), k  # replicated state identical on both ranks
    dev = torch.device("cuda:0")
    pts = synth.make_cloud(60_000, seed=9)
    poses, quats = synth.make_path(9, optical=True, jitter_seed=9)
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                  synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev)
    loss = m(vis_wps_dist=0.0)
    loss.backward()
    assert abs(loss.item() - float(r0["loss"])) <= 2e-6 * abs(loss.item())
    # the shard sums (a+b)+... in a different association than the single-process loop: 1e-6-level differences
    np.testing.assert_allclose(r0["rewards"], m.rewards.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    pg, qg = m.poses.grad.cpu().numpy(), m.quats.grad.cpu().numpy()
    assert np.abs(r0["pg"] - pg).max() <= 2e-5 * np.abs(pg).max()
    assert np.abs(r0["qg"] - qg).max() <= 2e-5 * np.abs(qg).max()
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    m2 = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                   synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev)
    res = optimize_trajectory(m2, n_opt_steps=4, lr_pose=0.05, lr_quat=0.01, rewards_th=1e9, vis_wps_dist=0.0)
    np.testing.assert_allclose(r0["opt_losses"], res.losses, rtol=2e-5)
    np.testing.assert_allclose(r0["opt_poses"], m2.poses.detach().cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(r0["opt_quats"], m2.quats.detach().cpu().numpy(), atol=2e-4)
