"""World-size-2 gloo test of the waypoint sharding + collectives (no GPU): the placement logic of
trajectory_optimization_amd.distributed with the CPU oracle standing in for the local kernels.
Property: sharded result == single-process result (log-odds are additive over waypoint shards)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pts, n_wps, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="2" if world <= 2 else "1")
    from trajectory_optimization_amd import synth
    from trajectory_optimization_amd.distributed import WaypointShard, init_from_env
    from oracle import oracle
    r, w, device = init_from_env(backend="gloo", use_gpu=False)
    assert (r, w, device.type) == (rank, world, "cpu")
    shard = WaypointShard()
    K, iw, ih = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
    pts = synth.make_cloud(n_pts, seed=5)
    poses, quats = synth.make_path(n_wps, optical=True, jitter_seed=5)
    lo, hi = shard.bounds(n_wps)
    # local forward (stand-in for tohip_traj_forward): partial log-odds of this rank's waypoints (a rank without any: zeros)
    if hi > lo:
        f = oracle.traj_forward(pts, poses[lo:hi], quats[lo:hi], K, iw, ih, prec="f64")
        lo_sum = torch.from_numpy(f["lo_sum"].copy())
    else:
        lo_sum = torch.zeros(n_pts, dtype=torch.float64)
    pending = shard.allreduce_sum_async(lo_sum)                                     # the one data-path collective, started ...
    overlapped = float(np.square(poses[lo:hi]).sum())                               # ... independent work in between (bench: the scan)
    pending.wait()
    assert overlapped >= 0.0
    rewards = 1.0 / (1.0 + torch.exp(-lo_sum))
    mean = rewards.mean().item()
    fwd = dict(rewards=rewards.numpy(), mean_reward=mean)
    pg = torch.zeros(n_wps, 3, dtype=torch.float64)
    qg = torch.zeros(n_wps, 4, dtype=torch.float64)
    pg_l, qg_l = np.zeros((0, 3)), np.zeros((0, 4))
    if hi > lo:
        pg_l, qg_l = oracle.traj_backward(pts, poses[lo:hi], quats[lo:hi], K, iw, ih, fwd, prec="f64")
        pg[lo:hi], qg[lo:hi] = torch.from_numpy(pg_l), torch.from_numpy(qg_l)
    pg, qg = shard.allreduce_sum(pg), shard.allreduce_sum(qg)                       # (W,7) gradient assembly
    gathered = None
    if n_wps % world == 0:  # equal shards (bench.py's weak-scaling layout): rows assembled by one all-gather instead
        gathered = shard.allgather_rows(torch.from_numpy(np.concatenate([pg_l, qg_l], axis=1)))
        assert torch.equal(gathered, torch.cat([pg, qg], dim=1))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rewards=rewards.numpy(), pg=pg.numpy(), qg=qg.numpy(),
             bounds=np.array([lo, hi]), gathered=np.zeros(0) if gathered is None else gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_wps", [7, 8])
def test_waypoint_sharding_world2(tmp_path, n_wps):
    sys.path.insert(0, REPO)
    from trajectory_optimization_amd import synth
    from oracle import oracle
    n_pts, world = 5000, 2
    mp.spawn(_worker, args=(world, _free_port(), n_pts, n_wps, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    # contiguous balanced cover of the waypoints
    assert r0["bounds"][0] == 0 and r0["bounds"][1] == r1["bounds"][0] and r1["bounds"][1] == n_wps
    assert abs((r0["bounds"][1] - r0["bounds"][0]) - (r1["bounds"][1] - r1["bounds"][0])) <= 1
    # replicated state is identical on both ranks
    for k in ("rewards", "pg", "qg"):
        assert np.array_equal(r0[k], r1[k])
    # and equals the single-process evaluation
    K, iw, ih = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
    pts = synth.make_cloud(n_pts, seed=5)
    poses, quats = synth.make_path(n_wps, optical=True, jitter_seed=5)
    f = oracle.traj_forward(pts, poses, quats, K, iw, ih, prec="f64")
    pg, qg = oracle.traj_backward(pts, poses, quats, K, iw, ih, f, prec="f64")
    np.testing.assert_allclose(r0["rewards"], f["rewards"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r0["pg"], pg, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(r0["qg"], qg, rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize("n_wps", [11, 5, 16])
def test_waypoint_sharding_world8(tmp_path, n_wps):
    """The driver's N = 8 layout rehearsed on the CPU (gloo, the oracle as the local compute): eight ranks, waypoint counts that do
    not divide (11: ranks of two and of one), fewer waypoints than ranks (5: three ranks hold none and still take part in every
    collective) and the even case (16: bench.py's all-gather of equal row blocks)."""
    sys.path.insert(0, REPO)
    from trajectory_optimization_amd import synth
    from oracle import oracle
    n_pts, world = 3000, 8
    mp.spawn(_worker, args=(world, _free_port(), n_pts, n_wps, str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert rs[0]["bounds"][0] == 0 and rs[-1]["bounds"][1] == n_wps and all(rs[r]["bounds"][1] == rs[r + 1]["bounds"][0] for r in range(world - 1))
    sizes = [int(r["bounds"][1] - r["bounds"][0]) for r in rs]
    assert max(sizes) - min(sizes) <= 1 and (n_wps >= world or min(sizes) == 0)
    for r in rs[1:]:
        for k in ("rewards", "pg", "qg"):
            assert np.array_equal(rs[0][k], r[k]), k
    K, iw, ih = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
    pts = synth.make_cloud(n_pts, seed=5)
    poses, quats = synth.make_path(n_wps, optical=True, jitter_seed=5)
    f = oracle.traj_forward(pts, poses, quats, K, iw, ih, prec="f64")
    pg, qg = oracle.traj_backward(pts, poses, quats, K, iw, ih, f, prec="f64")
    np.testing.assert_allclose(rs[0]["rewards"], f["rewards"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(rs[0]["pg"], pg, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(rs[0]["qg"], qg, rtol=1e-9, atol=1e-15)


def _worker_points(rank, world, port, n_pts, n_wps, out_dir):
    """distributed.PointShard's placement and collectives with the CPU oracle as the local compute: this rank's rows of the cloud,
    every waypoint; MAX of the waypoints' extrema in the int32 words the kernels use (-bits(min), bits(max), 0, 0); SUM of the
    40 doubles per waypoint and of the reward sum."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="2" if world <= 2 else "1")
    from trajectory_optimization_amd import synth
    from trajectory_optimization_amd.distributed import PointShard, init_from_env
    from oracle import oracle
    init_from_env(backend="gloo", use_gpu=False)
    shard = PointShard()
    assert shard.bounds(n_wps) == (0, n_wps)
    K, iw, ih = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
    pts = synth.make_cloud(n_pts, seed=5)
    poses, quats = synth.make_path(n_wps, optical=True, jitter_seed=5)
    lo, hi = shard.point_bounds(n_pts)
    mine = pts[lo:hi]
    # pass 1 on this rank's points (stand-in for tohip_traj_pshard_pass1), then collective 1 in the kernels' own representation:
    # p >= 0, so the int32 bit patterns of the f32 extrema order like the values and one MAX combines both fields
    pmin, pmax = oracle.traj_extrema(mine, poses, quats, K, iw, ih, prec="f32")
    words = np.zeros((n_wps, 4), np.int32)
    words[:, 0] = -pmin.astype(np.float32).view(np.int32)
    words[:, 1] = pmax.astype(np.float32).view(np.int32)
    t = torch.from_numpy(words.reshape(-1).copy())
    shard.allreduce_max(t)
    words = t.numpy().reshape(n_wps, 4)
    gmin32, gmax32 = (-words[:, 0]).astype(np.int32).view(np.float32), words[:, 1].copy().view(np.float32)
    # (the kernels work in f32 throughout; this rehearsal's arithmetic is the f64 oracle's, whose tie rules compare p with the
    # extrema by ==: the extrema it goes on with are the f64 ones, combined the same way)
    pmin, pmax = oracle.traj_extrema(mine, poses, quats, K, iw, ih, prec="f64")
    t64 = torch.from_numpy(np.concatenate([-pmin, pmax]))
    shard.allreduce_max(t64)
    gmin, gmax = -t64.numpy()[:n_wps], t64.numpy()[n_wps:].copy()
    # flags / log-odds / rewards of this rank's points against the global extrema, the 40 sums per waypoint with unit upstream
    # gradient (stand-in for tohip_traj_pshard_local), collective 2
    lo_sum, rewards = oracle.traj_forward_ext(mine, poses, quats, K, iw, ih, gmin, gmax, prec="f64")
    partial = np.zeros(8 + 40 * n_wps)
    partial[0], partial[2] = rewards.sum(), len(mine)
    partial[8:] = oracle.traj_backward_partial(mine, poses, quats, K, iw, ih, gmin, gmax, rewards, prec="f64").reshape(-1)
    tp = torch.from_numpy(partial)
    shard.allreduce_sum(tp)
    partial = tp.numpy()
    # the finish, identical on every rank (stand-in for tohip_traj_pshard_finish)
    mean = partial[0] / partial[2]
    vis = 1.0 / (mean + 1e-6)
    pg, qg = oracle.traj_backward_final(poses, quats, partial[8:].reshape(n_wps, 40), -vis * vis / partial[2], prec="f64")
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rewards=rewards, bounds=np.array([lo, hi]), pg=pg, qg=qg, vis=vis, gmin=gmin32, gmax=gmax32,
             gmin64=gmin, gmax64=gmax, n_all=partial[2])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pts", [5000, 5001])
def test_point_sharding_world2(tmp_path, n_pts):
    """Property: a point-sharded evaluation == the single-process one.  Maxima and minima do not depend on how the points are
    split (the combined extrema are the single process's f32 values to the bit); log-odds sums are complete per point on its own
    rank; everything after the gradient sums is linear in them."""
    sys.path.insert(0, REPO)
    from trajectory_optimization_amd import synth
    from oracle import oracle
    n_wps, world = 6, 2
    mp.spawn(_worker_points, args=(world, _free_port(), n_pts, n_wps, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    assert r0["bounds"][0] == 0 and r0["bounds"][1] == r1["bounds"][0] and r1["bounds"][1] == n_pts and int(r0["n_all"]) == n_pts
    for k in ("pg", "qg", "vis", "gmin", "gmax"):
        assert np.array_equal(r0[k], r1[k]), k   # replicated on both ranks
    K, iw, ih = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
    pts = synth.make_cloud(n_pts, seed=5)
    poses, quats = synth.make_path(n_wps, optical=True, jitter_seed=5)
    smin, smax = oracle.traj_extrema(pts, poses, quats, K, iw, ih, prec="f32")
    assert np.array_equal(r0["gmin"], smin.astype(np.float32)) and np.array_equal(r0["gmax"], smax.astype(np.float32))
    smin, smax = oracle.traj_extrema(pts, poses, quats, K, iw, ih, prec="f64")
    assert np.array_equal(r0["gmin64"], smin) and np.array_equal(r0["gmax64"], smax)
    # the single-process evaluation, cut at the same places
    lo_sum, rewards = oracle.traj_forward_ext(pts, poses, quats, K, iw, ih, smin, smax, prec="f64")
    assert np.array_equal(np.concatenate([r0["rewards"], r1["rewards"]]), rewards)
    vis = 1.0 / (rewards.mean() + 1e-6)
    part = oracle.traj_backward_partial(pts, poses, quats, K, iw, ih, smin, smax, rewards, prec="f64")
    pg, qg = oracle.traj_backward_final(poses, quats, part, -vis * vis / n_pts, prec="f64")
    assert abs(float(r0["vis"]) - vis) <= 1e-13 * vis
    np.testing.assert_allclose(r0["pg"], pg, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(r0["qg"], qg, rtol=1e-9, atol=1e-15)
    # and it is the model's gradient: the plain oracle, uncut
    f = oracle.traj_forward(pts, poses, quats, K, iw, ih, prec="f64")
    pg0, qg0 = oracle.traj_backward(pts, poses, quats, K, iw, ih, f, prec="f64")
    np.testing.assert_allclose(rewards, f["rewards"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r0["pg"], pg0, rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(r0["qg"], qg0, rtol=1e-8, atol=1e-14)


def test_bounds_cover_all_ranks():
    sys.path.insert(0, REPO)
    from trajectory_optimization_amd.distributed import WaypointShard
    s = WaypointShard.__new__(WaypointShard)
    for world in (1, 2, 3, 8):
        s.world_size = world
        for n in (0, 1, 5, 8, 127, 1024):
            prev = 0
            for r in range(world):
                s.rank = r
                lo, hi = s.bounds(n)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n


def test_point_sharding_world8(tmp_path):
    """Eight ranks, a point count that does not divide (3 005): the all-reduced extrema are the single process's to the bit, the
    rewards concatenate to the single process's, the gradients agree."""
    sys.path.insert(0, REPO)
    from trajectory_optimization_amd import synth
    from oracle import oracle
    n_pts, n_wps, world = 3005, 5, 8
    mp.spawn(_worker_points, args=(world, _free_port(), n_pts, n_wps, str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert rs[0]["bounds"][0] == 0 and rs[-1]["bounds"][1] == n_pts and all(rs[r]["bounds"][1] == rs[r + 1]["bounds"][0] for r in range(world - 1))
    for r in rs[1:]:
        for k in ("pg", "qg", "vis", "gmin", "gmax"):
            assert np.array_equal(rs[0][k], r[k]), k
    K, iw, ih = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
    pts = synth.make_cloud(n_pts, seed=5)
    poses, quats = synth.make_path(n_wps, optical=True, jitter_seed=5)
    smin, smax = oracle.traj_extrema(pts, poses, quats, K, iw, ih, prec="f32")
    assert np.array_equal(rs[0]["gmin"], smin.astype(np.float32)) and np.array_equal(rs[0]["gmax"], smax.astype(np.float32))
    f = oracle.traj_forward(pts, poses, quats, K, iw, ih, prec="f64")
    pg0, qg0 = oracle.traj_backward(pts, poses, quats, K, iw, ih, f, prec="f64")
    np.testing.assert_allclose(np.concatenate([r["rewards"] for r in rs]), f["rewards"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(rs[0]["pg"], pg0, rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(rs[0]["qg"], qg0, rtol=1e-8, atol=1e-14)
