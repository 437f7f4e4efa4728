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
    # the compact all-reduce (a bit per slot OR-reduced, then only the union's slots summed) gives the full one's bits
    mc = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                   synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device, shard=WaypointShard(compact=True))
    lc = mc(vis_wps_dist=0.0)
    lc.backward()
    compact_same = bool(torch.equal(lc.detach(), loss.detach()) and torch.equal(mc.rewards.detach(), m.rewards.detach()) and
                        torch.equal(mc.poses.grad, m.poses.grad) and torch.equal(mc.quats.grad, m.quats.grad))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=loss.item(), rewards=m.rewards.detach().cpu().numpy(),
             pg=m.poses.grad.cpu().numpy(), qg=m.quats.grad.cpu().numpy(), opt_poses=m2.poses.detach().cpu().numpy(),
             opt_quats=m2.quats.detach().cpu().numpy(), opt_losses=np.asarray(res.losses), compact_same=compact_same)
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
    assert bool(r0["compact_same"]) and bool(r1["compact_same"])
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


# ---- point sharding (distributed.PointShard): every rank a part of the cloud, all the waypoints ------------------------------

def _worker_points(rank, world, port, out_dir, vwd, rig):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from trajectory_optimization_amd import synth
    from trajectory_optimization_amd.distributed import PointShard, init_from_env
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    _, _, device = init_from_env(backend="gloo")
    pts = synth.make_cloud(70_001, seed=19)   # (an odd count: the ranks' parts differ in size)
    poses, quats = synth.make_path(11, optical=True, jitter_seed=19)
    kw = dict(rig=synth.camera_rig(3)) if rig else {}

    def model():
        return ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                         synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device, shard=PointShard(), **kw)
    m = model()
    lo_p, hi_p = m._shard.point_bounds(len(pts))
    assert m.points.shape[0] == hi_p - lo_p and m._n_global == len(pts)
    loss = m(vis_wps_dist=vwd)
    loss.backward()
    m2 = model()
    res = optimize_trajectory(m2, n_opt_steps=4, lr_pose=0.05, lr_quat=0.01, rewards_th=1e9, vis_wps_dist=vwd)
    ext = m._point_step((len(poses) + m._wps_step(vwd) - 1) // m._wps_step(vwd)).extrema.cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=loss.item(), vis=float(m.loss["vis"]), rewards=m.rewards.detach().cpu().numpy(),
             bounds=np.array([lo_p, hi_p]), pg=m.poses.grad.cpu().numpy(), qg=m.quats.grad.cpu().numpy(), extrema=ext,
             opt_poses=m2.poses.detach().cpu().numpy(), opt_quats=m2.quats.detach().cpu().numpy(), opt_losses=np.asarray(res.losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("vwd,rig", [(0.0, False), (2.5, False), (0.0, True)])
def test_point_sharded_model_equals_single_process(tmp_path, vwd, rig):
    """Two gloo ranks on the one GPU, each with half of the cloud and ALL the waypoints (PointShard): the extrema all-reduced
    after pass 1 are the single process's to the bit (a maximum does not depend on how the points are split), rewards within
    1e-6, loss and gradients (identical on both ranks) within 1e-5 of the single-process model; the launch-only optimiser follows."""
    sys.path.insert(0, REPO)
    from trajectory_optimization_amd import ops, synth
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    world = 2
    mp.spawn(_worker_points, args=(world, _free_port(), str(tmp_path), vwd, rig), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    for k in ("loss", "vis", "pg", "qg", "extrema", "opt_poses", "opt_quats", "opt_losses"):
        assert np.array_equal(r0[k], r1[k]), k   # replicated state identical on both ranks
    dev = torch.device("cuda:0")
    pts = synth.make_cloud(70_001, seed=19)
    poses, quats = synth.make_path(11, optical=True, jitter_seed=19)
    kw = dict(rig=synth.camera_rig(3)) if rig else {}
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                  synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev, **kw)
    loss = m(vis_wps_dist=vwd)
    loss.backward()
    # the waypoints' extrema: bitwise the single process's (its workspace holds them in the same words)
    step_w = m._wps_step(vwd)
    n_eval = (len(poses) + step_w - 1) // step_w
    C = 3 if rig else 1
    words, n_words = __import__("ctypes").c_void_p(), __import__("ctypes").c_int64()
    from trajectory_optimization_amd import _lib
    ws = m._workspace(n_eval)
    _lib.check(_lib.lib().tohip_traj_extrema_view(m._cloud.n, n_eval * C, _lib.ptr(ws.buf), ws.bytes, __import__("ctypes").byref(words),
                                                  __import__("ctypes").byref(n_words)), "extrema view")
    off = words.value - ws.buf.data_ptr()
    single_ext = ws.buf[off:off + 4 * n_words.value].view(torch.int32).cpu().numpy()
    assert np.array_equal(single_ext, r0["extrema"])
    rewards = m.rewards.detach().cpu().numpy()
    for r in (r0, r1):
        lo_p, hi_p = r["bounds"]
        np.testing.assert_allclose(r["rewards"], rewards[lo_p:hi_p], rtol=1e-6, atol=0)
    assert abs(float(r0["loss"]) - loss.item()) <= 2e-6 * abs(loss.item())
    pg, qg = m.poses.grad.cpu().numpy(), m.quats.grad.cpu().numpy()
    assert np.abs(r0["pg"] - pg).max() <= 1e-5 * np.abs(pg).max()
    assert np.abs(r0["qg"] - qg).max() <= 1e-5 * np.abs(qg).max()
    m2 = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                   synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev, **kw)
    res = optimize_trajectory(m2, n_opt_steps=4, lr_pose=0.05, lr_quat=0.01, rewards_th=1e9, vis_wps_dist=vwd)
    np.testing.assert_allclose(r0["opt_losses"], res.losses, rtol=2e-5)
    np.testing.assert_allclose(r0["opt_poses"], m2.poses.detach().cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(r0["opt_quats"], m2.quats.detach().cpu().numpy(), atol=2e-4)


# ---- bench.py's own N>1 step (the code the driver's 8-GPU run executes), rehearsed on the one GPU of the test box ------------

def _run(cmd, env_extra, timeout=600):
    import subprocess
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + "\n---\n" + r.stderr[-3000:]
    return r.stdout


_BENCH_COMMON = ["--points", "200000", "--steps", "2", "--warmup", "1", "--cpu-wps", "0", "--mode", "dense"]


@pytest.fixture(scope="module")
def bench_one_rank(tmp_path_factory):
    import json
    d = tmp_path_factory.mktemp("bench")
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--wps-per-gpu", "64", "--dump", str(d / "n1.npz")] + _BENCH_COMMON, {})
    last = out.strip().splitlines()[-1]
    assert len(last) < 8000      # the driver keeps an 8 KB tail of stdout: the headline must fit it whole
    line = json.loads(last)
    assert line["n_gpus"] == 1 and line["roofline"]["bound"] == "valu" and 0.0 < line["roofline"]["frac"] <= 1.0
    return np.load(d / "n1.npz")


def test_bench_step_two_ranks_equal_one_rank(bench_one_rank, tmp_path):
    """`python bench.py --gpus 2` as the driver types it, NO launcher around it: bench.py starts `torch.distributed.run
    --nproc-per-node 2` itself (before anything touches the GPU), relays rank 0's headline as its own last line and exits with the
    ranks' status.  gloo rendezvous, both ranks on the one GPU, host-staged collectives: waypoint shards of 32 + 32, ONE all-reduce
    of the log-odds vector, all-gather of the (W,7) gradient rows — the step's outputs equal the single-rank run over the same 64
    waypoints."""
    import json
    env = {"TOHIP_DIST_BACKEND": "gloo"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        assert k not in os.environ, k
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--wps-per-gpu", "32", "--dump", str(tmp_path / "n2.npz")] + _BENCH_COMMON, env)
    last = out.strip().splitlines()[-1]
    assert len(last) < 8000
    line = json.loads(last)
    assert line["n_gpus"] == 2 and line["config"]["waypoints_total"] == 64 and line["scaling"] == "weak"
    assert line["ranks_seen"] == 2 and line["backend"] == "gloo" and "allreduce_ms_median" in line["comm"]
    a, b = bench_one_rank, np.load(tmp_path / "n2.npz")
    # the two shards' log-odds are added in a different association than the single run's: 1e-6-level differences
    np.testing.assert_allclose(b["rewards"], a["rewards"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(b["scalars"][:2], a["scalars"][:2], rtol=2e-6)
    assert b["pg"].shape == a["pg"].shape == (64, 3) and b["qg"].shape == (64, 4)
    assert np.abs(b["pg"] - a["pg"]).max() <= 2e-5 * np.abs(a["pg"]).max()
    assert np.abs(b["qg"] - a["qg"]).max() <= 2e-5 * np.abs(a["qg"]).max()
    # a launcher around it (the driver's N > 1 command) still works: the ranks find WORLD_SIZE and do not spawn again
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--wps-per-gpu", "32"] + _BENCH_COMMON, env)
    line2 = json.loads([l for l in out.strip().splitlines() if l.startswith('{"metric"')][-1])
    assert line2["ranks_seen"] == 2 and line2["config"]["loss_vis"] == line["config"]["loss_vis"]


def test_bench_step_through_rccl_on_one_rank(bench_one_rank, tmp_path):
    """The same step with a ONE-rank `nccl` process group (TOHIP_DIST_FORCE_INIT): init_process_group(device_id=...), the
    all-reduce of the log-odds vector and the all-gather of the gradient rows all go through RCCL on the GPU — the calls the
    8-GPU run makes — and the outputs are the single-process ones to the bit."""
    import json
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--wps-per-gpu", "64", "--dump", str(tmp_path / "rccl.npz")] + _BENCH_COMMON,
               {"TOHIP_DIST_FORCE_INIT": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port())})
    line = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1
    a, b = bench_one_rank, np.load(tmp_path / "rccl.npz")
    for k in ("scalars", "pg", "qg", "rewards"):
        assert np.array_equal(a[k], b[k]), k
    # and with the compact all-reduce (OR-reduce of the slot mask + sum of the union's slots, both through RCCL)
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--wps-per-gpu", "64", "--compact-allreduce", "on", "--dump", str(tmp_path / "rcclc.npz")]
               + _BENCH_COMMON,
               {"TOHIP_DIST_FORCE_INIT": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port())})
    line = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert line["comm"]["compact_allreduce"] is True
    c = np.load(tmp_path / "rcclc.npz")
    for k in ("scalars", "pg", "qg", "rewards"):
        assert np.array_equal(a[k], c[k]), k


def test_bench_point_shard_two_ranks_and_one_rank_rccl(bench_one_rank, tmp_path):
    """bench.py --shard points: every rank a part of the cloud and all the waypoints.  Two gloo ranks on the one GPU, and a ONE-rank
    nccl group (the MAX all-reduce of the extrema words and the SUM all-reduce of the partial sums go through RCCL): loss and
    gradients equal the single-rank run's within the tolerance of a different summation order (f64 partials across ranks)."""
    import json
    a = bench_one_rank
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--wps-per-gpu", "32", "--shard", "points",
                "--dump", str(tmp_path / "p2.npz")] + _BENCH_COMMON, {"TOHIP_DIST_BACKEND": "gloo"})
    line = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["waypoints_total"] == 64 and line["config"]["parallelism"] == "point-shard x2"
    assert line["comm"]["shard"] == "points" and line["comm"]["extrema_allreduce_max"]["bytes"] == 64 * 16
    b = np.load(tmp_path / "p2.npz")
    np.testing.assert_allclose(b["scalars"][:2], a["scalars"][:2], rtol=2e-6)
    np.testing.assert_allclose(b["rewards"], a["rewards"][:len(b["rewards"])], rtol=1e-6, atol=0)   # rank 0's rows
    assert np.abs(b["pg"] - a["pg"]).max() <= 1e-5 * np.abs(a["pg"]).max()
    assert np.abs(b["qg"] - a["qg"]).max() <= 1e-5 * np.abs(a["qg"]).max()
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--wps-per-gpu", "64", "--shard", "points", "--dump", str(tmp_path / "p1.npz")] + _BENCH_COMMON,
               {"TOHIP_DIST_FORCE_INIT": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port())})
    line = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert line["comm"]["backend"] == "nccl" and line["comm"]["shard"] == "points"
    c = np.load(tmp_path / "p1.npz")
    assert np.array_equal(c["rewards"], a["rewards"]) and np.array_equal(c["scalars"][:2], a["scalars"][:2])
    assert np.abs(c["pg"] - a["pg"]).max() <= 1e-6 * np.abs(a["pg"]).max()
    assert np.abs(c["qg"] - a["qg"]).max() <= 1e-6 * np.abs(a["qg"]).max()


# ---- the reference's early stop under point sharding: the replicated mean reward (model.mean_reward) -------------------------

def _worker_points_early_stop(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from trajectory_optimization_amd import synth
    from trajectory_optimization_amd.distributed import PointShard, init_from_env
    from trajectory_optimization_amd.model import ModelTraj
    _, _, device = init_from_env(backend="gloo")
    # a cloud whose two halves see very different rewards: rows sorted along x, the path over the first half
    pts = synth.make_cloud(60_000, seed=23)
    pts = pts[np.argsort(pts[:, 0])]
    poses, quats = synth.make_path(9, optical=True, jitter_seed=23)
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS),
                  synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device, shard=PointShard())
    opt = torch.optim.Adam([{"params": [m.poses], "lr": 0.05}, {"params": [m.quats], "lr": 0.01}])
    # TrajOpt.run (/root/reference/src/trajectory_optimization.py:100-127) with the replicated mean in place of torch.mean(model.rewards)
    rewards_th, smoothness_th, stop_at, local_gains, gains = 1.004, 0.5, -1, [], []
    reward0 = smooth0 = local0 = None
    for i in range(40):
        opt.zero_grad()
        loss = m(vis_wps_dist=0.0)
        loss.backward()
        opt.step()
        if i == 0:
            reward0, smooth0, local0 = m.mean_reward.clone(), m.loss["smooth"].detach().clone(), torch.mean(m.rewards.detach())
        vis_gain, smooth_gain = m.mean_reward / reward0, smooth0 / m.loss["smooth"].detach()
        gains.append(float(vis_gain))
        local_gains.append(float(torch.mean(m.rewards.detach()) / local0))
        if vis_gain > rewards_th and smooth_gain > smoothness_th:
            stop_at = i
            break
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), stop_at=stop_at, gains=np.asarray(gains), local_gains=np.asarray(local_gains),
             poses=m.poses.detach().cpu().numpy(), mean=float(m.mean_reward), local_mean=float(torch.mean(m.rewards.detach())))
    dist.barrier()
    dist.destroy_process_group()


def test_point_sharded_early_stop_uses_the_replicated_mean(tmp_path):
    """ModelTraj(shard=PointShard()).mean_reward: the reference's early-stop rule on it fires at the SAME step on every rank (on
    torch.mean(model.rewards) — this rank's rows only — the ranks would disagree and the slower one would wait in its next forward's
    collective for ever)."""
    world = 2
    mp.spawn(_worker_points_early_stop, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    assert int(r0["stop_at"]) == int(r1["stop_at"]) and 0 < int(r0["stop_at"]) < 39, (r0["stop_at"], r1["stop_at"])   # it fired, on both, together
    assert np.array_equal(r0["gains"], r1["gains"]) and np.array_equal(r0["poses"], r1["poses"]) and float(r0["mean"]) == float(r1["mean"])
    assert float(r0["local_mean"]) != float(r1["local_mean"])        # the rank-local means are different numbers ...
    # ... and their gains cross the threshold at different steps (or not at all): the hazard the replicated value removes
    first = [next((i for i, g in enumerate(r["local_gains"]) if g > 1.004), None) for r in (r0, r1)]
    assert first[0] != first[1] or not np.allclose(r0["local_gains"], r1["local_gains"]), first
