"""Edge cases of the HIP path vs the f64 oracle: tiny and ragged clouds, many waypoints on a small cloud,
general (non-pinhole) intrinsics, other clip limits, duplicated clouds (ties everywhere), degenerate inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_inf
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(dev, pts, poses, quats, Kmat=K, iw=IW, ih=IH, min_dist=1.0, max_dist=5.0, flags=0):
    from trajectory_optimization_amd import ops
    cloud = ops.PackedCloud(torch.from_numpy(np.ascontiguousarray(pts)).to(dev))
    cam = ops.Camera(Kmat, iw, ih, min_dist, max_dist)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    ws = ops.TrajWorkspace(cloud, p.shape[0])
    lo, mm = ops.traj_forward(cloud, p, q, cam, ws, flags=flags)
    r, sc = ops.traj_reward(cloud, lo, cam, ws)
    pg, qg = ops.traj_backward(cloud, p.shape[0], cam, ws, lo, scalars=sc, gout=torch.ones(1, device=dev), flags=flags)
    torch.cuda.synchronize()
    return r.cpu().numpy(), sc.cpu().numpy(), pg.cpu().numpy(), qg.cpu().numpy(), mm.cpu().numpy()


def _oracle(pts, poses, quats, Kmat=K, iw=IW, ih=IH, min_dist=1.0, max_dist=5.0):
    from oracle import oracle
    f = oracle.traj_forward(pts, poses, quats, Kmat, iw, ih, min_dist, max_dist, prec="f64")
    pg, qg = oracle.traj_backward(pts, poses, quats, Kmat, iw, ih, f, min_dist=min_dist, max_dist=max_dist, prec="f64")
    return f, pg, qg


def _near_cloud(n, seed):
    """points a few metres in front of a camera at the origin looking along +x (optical frame of synth.make_path)"""
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3)) * np.array([6.0, 8.0, 6.0]) + np.array([0.5, -6.0, -5.0])).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 65, 1023, 1024, 1025, 4097])
def test_tiny_and_ragged_clouds(dev, n):
    pts = _near_cloud(n, n)
    poses, quats = synth.make_path(3, optical=True, jitter_seed=n)
    poses = (poses * 0.05).astype(np.float32)  # keep the cameras next to the cloud
    for flags in (0, 1):
        r, sc, pg, qg, mm = _run(dev, pts, poses, quats, flags=flags)
        f, opg, oqg = _oracle(pts, poses, quats)
        if n == 1:
            # one point: max(p - min p) = 0 -> 0/0: the reference's rewards are NaN (model.py:226-227)
            assert np.isnan(f["rewards"]).all() and np.isnan(r).all()
            continue
        np.testing.assert_allclose(r, f["rewards"], rtol=2e-5, atol=2e-6)
        # min p is exp(-arg) with arg ~ 50: its f32 relative error is ~|arg| * 2^-23 (the reference's too)
        np.testing.assert_allclose(mm[:, 0], f["pmin"], rtol=2e-4, atol=1e-30)
        np.testing.assert_allclose(mm[:, 1], f["pmax"], rtol=1e-5, atol=1e-30)
        assert rel_inf(pg, opg) < 1e-5 and rel_inf(qg, oqg) < 1e-5


def test_many_waypoints_small_cloud(dev):
    pts = synth.make_cloud(3000, seed=77)
    poses, quats = synth.make_path(700, optical=True, jitter_seed=77)
    r, sc, pg, qg, _ = _run(dev, pts, poses, quats)
    f, opg, oqg = _oracle(pts, poses, quats)
    np.testing.assert_allclose(r, f["rewards"], rtol=5e-5, atol=5e-6)  # |lo_sum| up to 700 x 13.8
    assert rel_inf(pg, opg) < 1e-5 and rel_inf(qg, oqg) < 1e-5


def test_general_intrinsics_and_clip_limits(dev):
    """Non-pinhole K (skew, k22 != 1, non-zero bottom row) takes the general kernel variants; other clip limits
    move the Gaussian (model.py:20-21)."""
    Kg = np.array([[700.0, 3.5, 600.0], [1.25, 720.0, 500.0], [1e-3, -2e-3, 1.05]], dtype=np.float32)
    pts = synth.make_cloud(40_000, seed=5)
    poses, quats = synth.make_path(5, optical=True, jitter_seed=5)
    for flags in (0, 1):
        r, sc, pg, qg, _ = _run(dev, pts, poses, quats, Kmat=Kg, iw=1000.0, ih=900.0, min_dist=0.5, max_dist=8.0, flags=flags)
        f, opg, oqg = _oracle(pts, poses, quats, Kmat=Kg, iw=1000.0, ih=900.0, min_dist=0.5, max_dist=8.0)
        np.testing.assert_allclose(r, f["rewards"], rtol=2e-5, atol=2e-6)
        assert abs(sc[1] - f["loss_vis"]) <= 3e-6 * f["loss_vis"]
        assert rel_inf(pg, opg) < 1e-5 and rel_inf(qg, oqg) < 1e-5


def test_all_points_duplicated(dev):
    """Every point twice: every per-waypoint max (and min) is a tie; torch shares those gradients evenly."""
    base = _near_cloud(700, 3)
    pts = np.concatenate([base, base], axis=0)
    poses, quats = synth.make_path(4, optical=True, jitter_seed=3)
    poses = (poses * 0.05).astype(np.float32)
    for flags in (0, 1):
        r, sc, pg, qg, _ = _run(dev, pts, poses, quats, flags=flags)
        f, opg, oqg = _oracle(pts, poses, quats)
        np.testing.assert_allclose(r, f["rewards"], rtol=2e-5, atol=2e-6)
        assert rel_inf(pg, opg) < 1e-5 and rel_inf(qg, oqg) < 1e-5


def test_cloud_entirely_invisible(dev):
    """A cloud 200 m behind every camera: p underflows to 0 for every point, (p - 0)/0 = NaN in the reference."""
    pts = (synth.make_cloud(5000, seed=1) + np.array([-300.0, 0, 0], np.float32)).astype(np.float32)
    poses, quats = synth.make_path(3, optical=True)
    r, sc, pg, qg, mm = _run(dev, pts, poses, quats)
    f, _, _ = _oracle(pts, poses, quats)
    assert np.all(f["pmax"] == 0) and np.all(mm[:, 1] == 0)
    assert np.isnan(f["rewards"]).all() and np.isnan(r).all()


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize("n,row", [(5000, 1234), (300_000, 0), (300_000, 299_999)])
def test_a_nan_or_inf_coordinate_poisons_everything_like_the_reference(dev, bad, n, row):
    """One NaN / inf coordinate in the cloud: the reference's p is NaN for that point, torch.min() / max() propagate it, and every
    reward of every waypoint, the loss and every gradient entry are NaN (/root/reference/src/model.py:226-231; probed on the
    reference itself: 5 000 of 5 000 rewards, 12 + 16 of 12 + 16 gradient entries).  fmax / fmin and the culling would drop such a
    point silently: tohip_pack_cloud notes it, the probe makes every waypoint degenerate.  Both modes; ModelTraj and the raw calls;
    ModelPose: its sum, hence its loss."""
    from trajectory_optimization_amd.model import ModelPose, ModelTraj
    pts = synth.make_cloud(n, seed=5)
    pts[row, 1] = bad
    poses, quats = synth.make_path(4, optical=True, jitter_seed=5)
    f, opg, oqg = _oracle(pts, poses, quats)
    assert np.isnan(f["rewards"]).all() and np.isnan(f["loss_vis"]) and np.isnan(opg).all() and np.isnan(oqg).all()
    for flags in (0, 1):
        r, sc, pg, qg, _ = _run(dev, pts, poses, quats, flags=flags)
        assert np.isnan(r).all() and np.isnan(sc[0]) and np.isnan(sc[1])
        assert np.isnan(pg).all() and np.isnan(qg).all()
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev)
    loss = m(vis_wps_dist=0.0)
    loss.backward()
    assert torch.isnan(loss) and torch.isnan(m.rewards).all() and torch.isnan(m.poses.grad).all() and torch.isnan(m.quats.grad).all()
    mp = ModelPose(torch.from_numpy(pts), torch.from_numpy(poses[:1]), torch.from_numpy(quats[:1]), torch.from_numpy(K), IW, IH, device=dev)
    lp = mp()
    lp.backward()
    obs = mp.observations.detach().cpu().numpy()
    assert torch.isnan(lp) and np.isnan(obs[row]) and np.isfinite(np.delete(obs, row)).all() and torch.isnan(mp.trans.grad).all()


def test_pose_model_general_K(dev):
    from oracle import oracle
    from trajectory_optimization_amd.model import ModelPose
    Kg = np.array([[700.0, 3.5, 600.0], [1.25, 720.0, 500.0], [1e-3, -2e-3, 1.05]], dtype=np.float32)
    pts = synth.make_cloud(30_000, seed=8)
    t0 = np.array([[1.0, -2.0, 0.3]], np.float32)
    q0 = (synth.Q_OPTICAL[None] * 1.3).astype(np.float32)
    m = ModelPose(torch.from_numpy(pts), torch.from_numpy(t0), torch.from_numpy(q0), torch.from_numpy(Kg), 1000.0, 900.0,
                  min_dist=0.5, max_dist=8.0, device=dev)
    loss = m()
    loss.backward()
    obs, l = oracle.pose_forward(pts, t0, q0, Kg, 1000.0, 900.0, 0.5, 8.0, prec="f64")
    tg, qg = oracle.pose_backward(pts, t0, q0, Kg, 1000.0, 900.0, l, min_dist=0.5, max_dist=8.0, prec="f64")
    assert abs(loss.item() - l) <= 5e-6 * l
    np.testing.assert_allclose(m.observations.detach().cpu().numpy(), obs, rtol=5e-5, atol=1e-12)
    assert rel_inf(m.trans.grad.cpu().numpy(), tg) < 1e-5 and rel_inf(m.quat.grad.cpu().numpy(), qg) < 1e-5


def test_hard_path_empty_and_tiny(dev):
    from trajectory_optimization_amd import ops, _lib
    cam = ops.Camera(K, IW, IH)
    d, f, idx = ops.frustum_cull(torch.empty((3, 0), device=dev), cam, 1.0, 10.0)
    assert d.numel() == 0 and idx.numel() == 0
    with pytest.raises(_lib.HipError):
        ops.hidden_pts_removal(torch.rand(2, 3, device=dev))
    tet = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [1.0, 1.0, 1.0]], device=dev)
    from oracle import oracle
    idx, mask = ops.hidden_pts_removal(tet)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), oracle.hidden_pts_removal(tet.cpu().numpy())[0])
