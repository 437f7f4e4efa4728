"""The reference-shaped Python surface (ModelTraj / ModelPose + Adam loops) on the GPU vs golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_inf
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _traj_model(d, dev, **kw):
    from trajectory_optimization_amd.model import ModelTraj
    return ModelTraj(points=torch.from_numpy(d["points"]), wps_poses=torch.from_numpy(d["poses"]),
                     wps_quats=torch.from_numpy(d["quats"]), intrins=torch.from_numpy(K), img_width=IW, img_height=IH,
                     device=dev, **kw)


@pytest.mark.parametrize("name", ["traj_bundled_default", "traj_bundled_tilted_all", "traj_synth_10000x8", "traj_synth_clip"])
def test_model_traj_matches_reference(dev, name):
    d = load_golden(name)
    kw = {k: float(d[k]) for k in ("smoothness_weight", "traj_length_weight", "min_dist", "max_dist") if k in d}
    m = _traj_model(d, dev, **kw)
    loss = m(vis_wps_dist=float(d["vis_wps_dist"]))
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) <= 5e-6 * abs(float(d["loss"]))
    for k in ("vis", "l2", "smooth", "length"):
        assert abs(float(m.loss[k]) - float(d["loss_" + k])) <= 5e-6 * max(1.0, abs(float(d["loss_" + k])))
    np.testing.assert_allclose(m.rewards.detach().cpu().numpy(), d["rewards"], rtol=2e-5, atol=2e-6)
    # quats only receive gradient from the HIP visibility path: the north-star 1e-5 bar.  poses.grad also
    # carries the torch-side regularisers (criterion: arccos of nearly straight segments, evaluated in f32
    # by the reference and here in different op orders): 1e-4.
    assert rel_inf(m.quats.grad.cpu().numpy(), d["quats_grad"]) < 1e-5
    assert rel_inf(m.poses.grad.cpu().numpy(), d["poses_grad"]) < 1e-4
    if "vis_poses_grad" in d:
        m.zero_grad()
        m(vis_wps_dist=float(d["vis_wps_dist"]))
        m.loss["vis"].backward()
        assert rel_inf(m.poses.grad.cpu().numpy(), d["vis_poses_grad"]) < 1e-5
        assert rel_inf(m.quats.grad.cpu().numpy(), d["vis_quats_grad"]) < 1e-5


def test_known_answers(dev):
    """bundled cloud/path, identity quats, defaults."""
    d = load_golden("traj_bundled_default")
    m = _traj_model(d, dev)
    loss = m()
    assert m._wps_step(0.5) == 2
    assert abs(loss.item() - 6.954090595) < 3e-5
    assert abs(float(m.loss["vis"]) - 1.889989376) < 1e-5
    assert abs(float(m.loss["smooth"]) - 5.064101219) < 2e-5
    assert abs(m.rewards.mean().item() - 0.5291025) < 2e-6
    loss.backward()
    assert torch.all(m.quats.grad[1] == 0)  # not evaluated (wps_step 2): regularisers do not touch quats


def test_traj_adam_loop(dev):
    """/root/reference/src/trajectory_optimization.py:91-116 with the launch-file learning rates."""
    d = load_golden("traj_adam_bundled")
    d["points"] = load_golden("bundled")["pts"]
    m = _traj_model(d, dev)
    opt = torch.optim.Adam([{"params": [m.poses], "lr": float(d["lr_pose"])},
                            {"params": [m.quats], "lr": float(d["lr_quat"])}])
    losses = []
    for i in range(10):
        opt.zero_grad()
        loss = m()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if i + 1 in (1, 5, 10):
            # Adam's sign-like first steps amplify 1e-6 gradient noise where g ~ 0: absolute tolerance
            np.testing.assert_allclose(m.poses.detach().cpu().numpy(), d[f"poses_step{i + 1}"], rtol=0, atol=2e-3)
            np.testing.assert_allclose(m.quats.detach().cpu().numpy(), d[f"quats_step{i + 1}"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(losses, d["losses"], rtol=2e-3)


def _pose_model(d, dev):
    from trajectory_optimization_amd.model import ModelPose
    kw = {k: float(d[k]) for k in ("min_dist", "max_dist") if k in d}
    return ModelPose(points=torch.from_numpy(d["points"]), trans0=torch.from_numpy(d["trans0"]),
                     q0=torch.from_numpy(d["q0"]), intrins=torch.from_numpy(K), img_width=IW, img_height=IH, device=dev, **kw)


@pytest.mark.parametrize("name", ["pose_bundled_nohpr", "pose_bundled_tilted", "pose_synth_clip"])
def test_model_pose_matches_reference(dev, name):
    d = load_golden(name)
    m = _pose_model(d, dev)
    loss = m(hpr=bool(d["hpr"]))
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) <= 5e-6 * float(d["loss"])
    np.testing.assert_allclose(m.observations.detach().cpu().numpy(), d["observations"], rtol=5e-5, atol=1e-9)
    assert rel_inf(m.trans.grad.cpu().numpy(), d["trans_grad"]) < 1e-5
    assert rel_inf(m.quat.grad.cpu().numpy(), d["quat_grad"]) < 1e-5


def test_pose_that_sees_nothing(dev):
    """/root/reference/src/model.py:124-127 with every observation below FLT_MIN (fixture made by running the reference: loss 1/eps,
    54 subnormal observations summing to 1.3e-43, a gradient of ~1e-29): the hardware's exp flushes subnormals, so the HIP path's
    observations are zeros where the reference holds 1e-40s — the documented bound: the loss equal, every observation below
    FLT_MIN, |gradient| <= 1e-20 and finite."""
    d = load_golden("pose_sees_nothing")
    m = _pose_model(d, dev)
    loss = m(hpr=False)
    loss.backward()
    tiny = float(np.finfo(np.float32).tiny)
    assert abs(loss.item() - float(d["loss"])) <= 1e-6 * float(d["loss"])
    obs = m.observations.detach().cpu().numpy()
    assert float(obs.max()) < tiny and float(obs.min()) >= 0.0 and float(np.abs(obs - d["observations"]).max()) < tiny
    for g, ref in ((m.trans.grad, d["trans_grad"]), (m.quat.grad, d["quat_grad"])):
        g = g.cpu().numpy()
        assert np.isfinite(g).all() and np.abs(g).max() <= 1e-20 and np.abs(g - ref).max() <= 1e-20
    from trajectory_optimization_amd.optimizer import optimize_pose
    res = optimize_pose(_pose_model(d, dev), n_opt_steps=3)   # the launch-only loop: same loss, the pose stays finite
    assert all(abs(x - float(d["loss"])) <= 1e-6 * float(d["loss"]) for x in res.losses)


def test_pose_adam_loop(dev):
    d = load_golden("pose_adam_bundled")
    d["points"] = load_golden("bundled")["pts"]
    m = _pose_model(d, dev)
    opt = torch.optim.Adam([{"params": [m.trans], "lr": float(d["lr_pose"])},
                            {"params": [m.quat], "lr": float(d["lr_quat"])}])
    losses = []
    for i in range(10):
        opt.zero_grad()
        loss = m()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if i + 1 in (1, 5, 10):
            np.testing.assert_allclose(m.trans.detach().cpu().numpy(), d[f"trans_step{i + 1}"], rtol=0, atol=1e-3)
            np.testing.assert_allclose(m.quat.detach().cpu().numpy(), d[f"quat_step{i + 1}"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(losses, d["losses"], rtol=1e-3)


def test_device_resident_pose_optimizer(dev):
    """optimizer.optimize_pose (launch-only loop: pose forward/backward + HIP Adam) vs the reference's torch.optim.Adam run,
    and vs the drop-in path (model + torch.optim.Adam) step for step."""
    from trajectory_optimization_amd.optimizer import optimize_pose
    d = load_golden("pose_adam_bundled")
    d["points"] = load_golden("bundled")["pts"]
    for steps in (1, 5, 10):
        m = _pose_model(d, dev)
        res = optimize_pose(m, n_opt_steps=steps, lr_pose=float(d["lr_pose"]), lr_quat=float(d["lr_quat"]))
        np.testing.assert_allclose(m.trans.detach().cpu().numpy(), d[f"trans_step{steps}"], rtol=0, atol=1e-3)
        np.testing.assert_allclose(m.quat.detach().cpu().numpy(), d[f"quat_step{steps}"], rtol=0, atol=1e-3)
        np.testing.assert_allclose(res.losses, d["losses"][:steps], rtol=1e-3)
    # with the (pose-independent) world-frame HPR mask of model.py:114
    m1, m2 = _pose_model(d, dev), _pose_model(d, dev)
    res = optimize_pose(m1, n_opt_steps=4, lr_pose=0.05, lr_quat=0.02, hpr=True)
    opt = torch.optim.Adam([{"params": [m2.trans], "lr": 0.05}, {"params": [m2.quat], "lr": 0.02}])
    ref = []
    for _ in range(4):
        opt.zero_grad()
        loss = m2(hpr=True)
        loss.backward()
        opt.step()
        ref.append(loss.item())
    np.testing.assert_allclose(res.losses, ref, rtol=1e-5)
    np.testing.assert_allclose(m1.trans.detach().cpu().numpy(), m2.trans.detach().cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(m1.quat.detach().cpu().numpy(), m2.quat.detach().cpu().numpy(), atol=1e-5)


def test_errors_like_reference(dev):
    """W=1 -> the reference's int(NaN) ValueError; CPU device -> loud failure (no fallback)."""
    from trajectory_optimization_amd.model import ModelTraj
    pts = torch.from_numpy(synth.make_cloud(1000, seed=1))
    p, q = synth.make_path(1)
    m = ModelTraj(pts, torch.from_numpy(p), torch.from_numpy(q), torch.from_numpy(K), IW, IH, device=dev)
    with pytest.raises(ValueError):
        m()
    with pytest.raises(RuntimeError):
        ModelTraj(pts, torch.from_numpy(p), torch.from_numpy(q), torch.from_numpy(K), IW, IH, device=torch.device("cpu"))


def test_device_resident_optimizer_matches_reference_adam(dev):
    """optimizer.optimize_trajectory (launch-only loop: HIP regularisers + Adam + early stop) vs the reference's
    Adam trajectory on the bundled cloud/path (/root/reference/src/trajectory_optimization.py:91-116)."""
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    d = load_golden("traj_adam_bundled")
    d["points"] = load_golden("bundled")["pts"]
    for k in (1, 5, 10):
        m = _traj_model(d, dev)
        res = optimize_trajectory(m, n_opt_steps=k, lr_pose=float(d["lr_pose"]), lr_quat=float(d["lr_quat"]),
                                  rewards_th=1e9, smoothness_th=1e9)
        assert res.steps_taken == k and not res.stopped
        np.testing.assert_allclose(m.poses.detach().cpu().numpy(), d[f"poses_step{k}"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(m.quats.detach().cpu().numpy(), d[f"quats_step{k}"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(res.losses, d["losses"][:k], rtol=2e-3)


@pytest.mark.parametrize("vwd,rig", [(0.0, False), (1.5, False), (2.5, True)])
def test_one_call_step_equals_the_separate_calls(dev, vwd, rig):
    """optimize_trajectory's step as ONE call and five launches (tohip_traj_opt_step: strided waypoint reads, the regularisers and
    Adam's constants in the probe's launch, the update in the finish launch's blocks) against the separate calls of a sharded /
    occlusion-aware run (forward | reward + backward | tohip_traj_step_tail) — poses, quaternions, rewards, losses and the stop
    step to the bit."""
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd import optimizer as O
    pts = torch.from_numpy(synth.make_cloud(90_000, seed=31))
    p, q = synth.make_path(23, optical=True, jitter_seed=31)   # 23 waypoints: the last stride is ragged at step 2
    kw = dict(rig=synth.camera_rig(3)) if rig else {}
    runs = []
    for split in (False, True):
        m = ModelTraj(pts, torch.from_numpy(p), torch.from_numpy(q), torch.from_numpy(K), IW, IH, device=dev, **kw)
        args = (m, 7, 0.05, 0.01, 1.003, 0.5, vwd, (0.9, 0.999), 1e-8)
        res = O._optimize_trajectory_split(*args) if split else O.optimize_trajectory(m, *args[1:7])
        runs.append((m, res))
    (ma, ra), (mb, rb) = runs
    assert (ma._wps_step(vwd) > 1) == (vwd > 0.0)
    assert torch.equal(ma.poses.data, mb.poses.data) and torch.equal(ma.quats.data, mb.quats.data)
    assert torch.equal(ma.rewards, mb.rewards)
    assert (ra.steps_taken, ra.stopped, ra.losses) == (rb.steps_taken, rb.stopped, rb.losses)
    assert (ra.visibility_gain, ra.smoothness_gain) == (rb.visibility_gain, rb.smoothness_gain)
    for k in ("vis", "l2", "length", "smooth"):
        assert float(ma.loss[k]) == float(mb.loss[k])
    assert not torch.equal(ma.poses.data, ma.poses0)


def test_device_regularizers_match_torch(dev):
    """HIP regularisers (value + analytic gradient) vs the torch criterion of the model (autograd)."""
    from trajectory_optimization_amd import _lib
    from trajectory_optimization_amd._lib import ptr, stream_ptr, check
    d = load_golden("traj_bundled_tilted_all")
    m = _traj_model(d, dev, smoothness_weight=28.0, traj_length_weight=0.05)
    with torch.no_grad():
        m.poses.add_(0.05 * torch.randn(m.poses.shape, generator=torch.Generator().manual_seed(3)).to(dev))
    m.loss["vis"] = torch.zeros((), device=dev)
    reg = m.criterion(torch.full((10,), 0.5, device=dev))  # vis = 1/(0.5+eps): constant w.r.t. poses
    reg.backward()
    scal = torch.tensor([0.5, 1.0 / (0.5 + 1e-6), 0, 0], device=dev)
    lt = torch.zeros(8, device=dev)
    g = torch.zeros_like(m.poses)
    gt = torch.zeros((3,) + tuple(m.poses.shape), device=dev)
    check(_lib.lib().tohip_traj_regularizers(ptr(m.poses.data), ptr(m.poses0), m.poses.shape[0], 28.0, 0.05, 1e-6,
                                             ptr(scal), ptr(lt), ptr(g), 0, None, ptr(gt), stream_ptr()), "regularizers")
    torch.cuda.synchronize()
    np.testing.assert_allclose(gt.sum(0).cpu().numpy(), g.cpu().numpy(), rtol=1e-5, atol=1e-7)  # the per-term gradients add up
    assert abs(lt[4].item() - reg.item()) <= 2e-5 * abs(reg.item())
    for k, name in ((1, "l2"), (2, "length"), (3, "smooth")):
        assert abs(lt[k].item() - float(m.loss[name])) <= 2e-5 * max(1.0, abs(float(m.loss[name])))
    assert rel_inf(g.cpu().numpy(), m.poses.grad.cpu().numpy()) < 1e-4  # torch side is f32 arccos: 1e-5-level noise


def test_device_early_stop_rule(dev):
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    d = load_golden("traj_adam_bundled")
    d["points"] = load_golden("bundled")["pts"]
    m = _traj_model(d, dev)
    res = optimize_trajectory(m, n_opt_steps=12, lr_pose=0.12, lr_quat=0.05, rewards_th=1.02, smoothness_th=0.0)
    # the reference stops at the first step whose mean reward exceeds 1.02 x the initial one
    gains = np.array(d["mean_rewards"]) / d["mean_rewards"][0]
    first = int(np.argmax(gains > 1.02)) if (gains > 1.02).any() else None
    if first is not None:
        assert res.stopped and res.steps_taken == first + 1
    else:
        assert not res.stopped


@pytest.mark.parametrize("method", ["hpr", "zbuffer"])
def test_occlusion_aware_traj_model(dev, method):
    """ModelTraj(occlusion=...) multiplies each waypoint's p by a per-waypoint occlusion mask built by
    the hard pipeline of pc_processor.py (frustum cull -> HPR from the camera centre).  'hpr': masks, rewards and
    gradients vs the oracle running the same pipeline on the host; 'zbuffer': consistency with its own mask."""
    from oracle import oracle
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd import ops
    pts = synth.make_cloud(60_000, seed=21)
    poses, quats = synth.make_path(5, optical=True, jitter_seed=21)
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH,
                  device=dev, occlusion=method, occlusion_limits=(1.0, 15.0))
    m(vis_wps_dist=0.0)
    m.loss["vis"].backward()
    # the masks the model used, unpacked to the caller's point order
    rows = ops.occlusion_bits(m._cloud, m.points, m.poses.data, m.quats.data, m._cam, 1.0, 15.0, method).cpu().numpy()
    perm = m._cloud.perm.cpu().numpy()[:m._cloud.n]
    bits = ((rows.view(np.uint32)[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(rows.shape[0], -1)[:, :m._cloud.n]
    occ = np.zeros((rows.shape[0], m._cloud.n), np.float32)
    occ[:, perm] = bits
    if method == "hpr":
        ref_occ = oracle.occlusion_masks(pts, poses, quats, K, IW, IH, 1.0, 15.0)
        assert np.array_equal(occ, ref_occ)            # bit-exact hidden sets per waypoint
        assert 0 < (ref_occ == 0).sum() < ref_occ.size
    f = oracle.traj_forward(pts, poses, quats, K, IW, IH, prec="f64", occ=occ)
    pg, qg = oracle.traj_backward(pts, poses, quats, K, IW, IH, f, prec="f64")
    np.testing.assert_allclose(m.rewards.detach().cpu().numpy(), f["rewards"], rtol=2e-5, atol=2e-6)
    assert abs(float(m.loss["vis"].item()) - f["loss_vis"]) <= 3e-6 * f["loss_vis"]
    assert rel_inf(m.poses.grad.cpu().numpy(), pg) < 1e-5 and rel_inf(m.quats.grad.cpu().numpy(), qg) < 1e-5
    # occlusion can only lower a point's reward relative to the unoccluded model
    m0 = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev)
    m0(vis_wps_dist=0.0)
    assert m.rewards.mean().item() <= m0.rewards.mean().item() + 1e-7
    # dense evaluation of the same masks: same bits
    md = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH,
                   device=dev, occlusion=method, dense=True)
    md(vis_wps_dist=0.0)
    assert torch.equal(md.rewards, m.rewards)


def test_occlusion_refresh_policy(dev):
    """ModelTraj(occlusion_refresh_every=k): the masks are rebuilt on every k-th forward and reused in between (they are piecewise
    constant in the poses and carry no gradient).  k = 1 is the model without the policy, bit for bit; k = 3 holds the same rows
    object for three forwards; refresh_occlusion() forces a rebuild; the launch-only optimiser follows the same rule."""
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    pts = torch.from_numpy(synth.make_cloud(80_000, seed=33))
    poses, quats = synth.make_path(6, optical=True, jitter_seed=33)

    def model(k):
        return ModelTraj(pts, torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev, occlusion="zbuffer",
                         occlusion_refresh_every=k)

    def loop(m, n):
        opt = torch.optim.Adam([{"params": [m.poses], "lr": 0.1}, {"params": [m.quats], "lr": 0.02}])
        ids, losses = [], []
        for _ in range(n):
            opt.zero_grad()
            loss = m(vis_wps_dist=0.0)
            loss.backward()
            opt.step()
            ids.append(m._occ_cache[0].data_ptr())
            losses.append(loss.item())
        return ids, losses
    m1 = model(1)
    ids1, l1 = loop(m1, 6)
    m3 = model(3)
    ids3, l3 = loop(m3, 7)
    assert ids3[0] == ids3[1] == ids3[2] and ids3[3] == ids3[4] == ids3[5] and ids3[2] != ids3[3] and ids3[5] != ids3[6]
    assert l3[0] == l1[0]                       # the first forward builds the same masks
    assert abs(l3[5] - l1[5]) < 0.05 * l1[5]    # and reusing them for two more steps changes little (0.1 m per step)
    m3.refresh_occlusion()
    m3(vis_wps_dist=0.0)
    assert m3._occ_cache[2] == 1 and m3._occ_cache[0].data_ptr() != ids3[6]
    # the launch-only loop: 1 step with k = 1 == 1 step with k = 3 (same first masks); 4 steps differ only through the masks' age
    a, b = model(1), model(3)
    ra = optimize_trajectory(a, n_opt_steps=1, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=0.0)
    rb = optimize_trajectory(b, n_opt_steps=1, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=0.0)
    assert ra.losses == rb.losses and torch.equal(a.poses.data, b.poses.data)


def test_sample_script_and_npz_format(dev, tmp_path):
    """The ROS-free twin of trajectory_optimization_sample.py on files in the reference's .npz sample format
    ((3,N) point layout included): runs, improves visibility, writes normalised quaternions."""
    import importlib.util
    import os
    from trajectory_optimization_amd import samples
    b = load_golden("bundled")
    np.savez(tmp_path / "point_cloud_0.npz", pts=b["pts"].T)  # the (3,N) layout the sample transposes
    np.savez(tmp_path / "path_poses_0.npz", poses=b["poses"])
    pts, poses, quats = samples.load_data(tmp_path / "point_cloud_0.npz", tmp_path / "path_poses_0.npz")
    assert np.array_equal(pts, b["pts"]) and np.array_equal(poses, b["poses"]) and quats.shape == (len(poses), 4)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sample", os.path.join(repo, "examples", "trajectory_optimization_sample.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "res.npz"
    log = mod.main(["--points", str(tmp_path / "point_cloud_0.npz"), "--poses", str(tmp_path / "path_poses_0.npz"),
                    "--opt-steps", "20", "--out", str(out)])
    r = np.load(out)
    assert r["poses"].shape == poses.shape and r["rewards"].shape == (len(pts),)
    np.testing.assert_allclose(np.linalg.norm(r["quats_wxyz"], axis=1), 1.0, atol=1e-6)
    assert log["visibility"][-1] > 1.0  # Adam on the visibility loss raises the mean reward within 20 steps


def test_pose_sample_script(dev, tmp_path):
    import importlib.util
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pose_sample", os.path.join(repo, "examples", "pose_optimization_sample.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "pose.npz"
    losses = mod.main(["--opt-steps", "30", "--out", str(out)])
    r = np.load(out)
    assert r["trans"].shape == (1, 3) and abs(np.linalg.norm(r["quat_wxyz"]) - 1.0) < 1e-6
    assert r["observations"].shape == (len(load_golden("bundled")["pts"]),)
    assert min(losses) < losses[0]  # more of the cloud in view than at the start


@pytest.mark.parametrize("case", ["bundled_step2", "synth_all", "three_wps"])
def test_fused_loss_node_equals_op_by_op_criterion(dev, case):
    """ModelTraj.forward as one autograd node (visibility + criterion regularisers on the device) against the
    rewards node + torch criterion it replaces: same loss terms and gradients; a loss built on model.rewards
    still differentiates; a subclass that overrides criterion keeps the op-by-op path."""
    from trajectory_optimization_amd.model import ModelTraj
    if case == "bundled_step2":
        b = load_golden("bundled")
        pts, poses, vwd = b["pts"], b["poses"], 0.5
        quats = np.tile(np.array([[1, 0, 0, 0]], np.float32), (len(poses), 1))
    elif case == "synth_all":
        pts = synth.make_cloud(60_000, seed=4)
        poses, quats = synth.make_path(11, optical=True, jitter_seed=3)
        vwd = 0.0
    else:
        pts = synth.make_cloud(5000, seed=4)
        poses, quats = synth.make_path(3, optical=True, jitter_seed=1)
        vwd = 0.0

    def build(fused, cls=ModelTraj):
        m = cls(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev)
        m.fused_loss = fused
        return m

    a, b_ = build(True), build(False)
    la, lb = a(vis_wps_dist=vwd), b_(vis_wps_dist=vwd)
    la.backward()
    lb.backward()
    assert la.grad_fn.__class__.__name__.startswith("_TrajLoss") and not lb.grad_fn.__class__.__name__.startswith("_TrajLoss")
    assert abs(la.item() - lb.item()) <= 5e-6 * abs(lb.item())
    for k in ("vis", "l2", "length", "smooth"):
        assert abs(float(a.loss[k]) - float(b_.loss[k])) <= 5e-6 * max(1.0, abs(float(b_.loss[k]))), k
    assert torch.equal(a.rewards, b_.rewards)
    assert rel_inf(a.poses.grad.cpu().numpy(), b_.poses.grad.cpu().numpy()) < 1e-4
    assert rel_inf(a.quats.grad.cpu().numpy(), b_.quats.grad.cpu().numpy()) < 1e-5
    # a criterion of the caller's own on model.rewards (+ the model's loss): the general dL/d rewards path
    c, d = build(True), build(False)
    w = torch.linspace(0.5, 1.5, len(pts), device=dev)
    for m in (c, d):
        loss = m(vis_wps_dist=vwd)
        (2.0 * loss + (w * m.rewards).sum() / len(pts)).backward()
    assert rel_inf(c.poses.grad.cpu().numpy(), d.poses.grad.cpu().numpy()) < 1e-4
    assert rel_inf(c.quats.grad.cpu().numpy(), d.quats.grad.cpu().numpy()) < 1e-5

    class Mine(ModelTraj):
        def criterion(self, rewards):
            return 1.0 / (rewards.mean() + self.eps)

    e = build(True, Mine)
    le = e(vis_wps_dist=vwd)
    assert not le.grad_fn.__class__.__name__.startswith("_TrajLoss")
    assert abs(le.item() - float(a.loss["vis"])) <= 5e-6 * le.item()


def test_fused_loss_terms_are_differentiable(dev):
    """Every entry of model.loss keeps a grad_fn (the reference's are torch expressions): each term alone gives the
    gradient the op-by-op criterion gives for it."""
    from trajectory_optimization_amd.model import ModelTraj
    pts = synth.make_cloud(40_000, seed=6)
    poses, quats = synth.make_path(9, optical=True, jitter_seed=4)
    for term in ("vis", "l2", "length", "smooth"):
        grads = []
        for fused in (True, False):
            m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev)
            m.fused_loss = fused
            with torch.no_grad():
                m.poses += 0.05 * torch.randn(m.poses.shape, generator=torch.Generator().manual_seed(1)).to(dev)  # l2 > 0
            m(vis_wps_dist=0.0)
            (3.0 * m.loss[term]).backward()
            grads.append((m.poses.grad.clone(), None if m.quats.grad is None else m.quats.grad.clone()))
        (pa, qa), (pb, qb) = grads
        # the op-by-op side differentiates arccos in f32 (1e-4-level noise on the smoothness term); the kernel works in f64
        assert rel_inf(pa.cpu().numpy(), pb.cpu().numpy()) < (5e-4 if term == "smooth" else 1e-4), term
        if term == "vis":
            assert rel_inf(qa.cpu().numpy(), qb.cpu().numpy()) < 1e-5
        else:
            assert qa is None or float(qa.abs().max()) == 0.0


@pytest.mark.parametrize("hpr", [False, True])
def test_pose_fused_loss_node(dev, hpr):
    """ModelPose.forward as one autograd node vs the observations node + torch criterion; observations stay differentiable."""
    from trajectory_optimization_amd.model import ModelPose
    d = load_golden("pose_bundled_hpr")
    out = []
    for fused in (True, False):
        for custom in (False, True):
            m = ModelPose(points=torch.from_numpy(d["points"]), trans0=torch.from_numpy(d["trans0"]), q0=torch.from_numpy(d["q0"]),
                          intrins=torch.from_numpy(K), img_width=IW, img_height=IH, device=dev)
            m.fused_loss = fused
            loss = m(hpr=hpr)
            assert loss.grad_fn.__class__.__name__.startswith("_PoseLoss") == fused
            if custom:
                w = torch.linspace(0.0, 2.0, m.observations.numel(), device=dev)
                (5.0 * loss + 1e-4 * (w * m.observations).sum()).backward()
            else:
                loss.backward()
            out.append((loss.item(), m.observations.detach().clone(), m.trans.grad.clone(), m.quat.grad.clone()))
    for a, b in ((out[0], out[2]), (out[1], out[3])):
        assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0]) and torch.equal(a[1], b[1])
        assert rel_inf(a[2].cpu().numpy(), b[2].cpu().numpy()) < 1e-5 and rel_inf(a[3].cpu().numpy(), b[3].cpu().numpy()) < 1e-5


def test_fused_adam_optimizer_equals_torch_adam(dev):
    """optimizer.Adam (one launch per parameter) steps like torch.optim.Adam, with a scheduler on its groups."""
    from trajectory_optimization_amd.optimizer import Adam
    g = torch.Generator().manual_seed(0)
    p0, q0 = torch.randn(27, 3, generator=g), torch.randn(27, 4, generator=g)
    runs = []
    for cls in (torch.optim.Adam, Adam):
        p, q = torch.nn.Parameter(p0.clone().to(dev)), torch.nn.Parameter(q0.clone().to(dev))
        opt = cls([{"params": [p], "lr": 0.1}, {"params": [q], "lr": 0.02}])
        sched = torch.optim.lr_scheduler.ExponentialLR(optimizer=opt, gamma=0.9)
        gg = torch.Generator().manual_seed(1)
        for i in range(25):
            opt.zero_grad()
            p.grad = torch.randn(27, 3, generator=gg).to(dev) * (1.0 + i)
            q.grad = torch.randn(27, 4, generator=gg).to(dev) * 1e-3
            opt.step()
            if i % 5 == 0:
                sched.step()
        runs.append((p.detach().clone(), q.detach().clone()))
    # parameters of magnitude ~1 after 25 steps of size ~0.1: a few ulp (1.2e-7 each) of accumulated rounding
    np.testing.assert_allclose(runs[0][0].cpu().numpy(), runs[1][0].cpu().numpy(), rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(runs[0][1].cpu().numpy(), runs[1][1].cpu().numpy(), rtol=2e-6, atol=1e-6)


def test_dense_model_equals_the_default_model(dev):
    """ModelTraj(dense=True) evaluates every pair in pass 1; results equal the default (culled) model's to the bit,
    through the fused node and through the rewards node."""
    from trajectory_optimization_amd.model import ModelTraj
    pts = synth.make_cloud(150_000, seed=7)
    poses, quats = synth.make_path(10, optical=True, jitter_seed=5)
    out = []
    for dense in (False, True):
        for fused in (True, False):
            m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH,
                          device=dev, dense=dense)
            m.fused_loss = fused
            loss = m(vis_wps_dist=0.0)
            loss.backward()
            out.append((loss.detach().clone(), m.rewards.detach().clone(), m.poses.grad.clone(), m.quats.grad.clone()))
    for fused_idx in (0, 1):
        a, b = out[fused_idx], out[2 + fused_idx]
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_occlusion_with_a_rig_equals_explicit_virtual_waypoints(dev):
    """ModelTraj(rig=..., occlusion='hpr'): each (waypoint, camera) pair gets its own occlusion row, built at the camera's own
    pose.  Equal to a rig-less model whose waypoints are the cameras written out (same rows, same rewards)."""
    from trajectory_optimization_amd.model import ModelTraj
    pts = synth.make_cloud(50_000, seed=23)
    poses, quats = synth.make_path(4, optical=True, jitter_seed=23)
    rq, rt = synth.camera_rig(3)
    rt = rt + np.array([[0.1, 0.0, 0.2], [0.0, 0.15, 0.2], [-0.1, 0.0, 0.25]], dtype=np.float32)
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH,
                  device=dev, rig=(rq, rt), occlusion="hpr")
    loss = m(vis_wps_dist=0.0)
    loss.backward()
    assert bool(torch.isfinite(m.poses.grad).all()) and float(m.poses.grad.abs().max()) > 0
    # explicit virtual waypoints on the host (f64), unit quaternions
    qn = quats.astype(np.float64) / np.linalg.norm(quats.astype(np.float64), axis=1, keepdims=True)
    vq = synth.quat_mul(qn[:, None, :], rq[None].astype(np.float64)).reshape(-1, 4)
    w, x, y, z = qn.T
    R = np.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                  w * w - x * x + y * y - z * z, 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                  w * w - x * x - y * y + z * z], -1).reshape(-1, 3, 3)
    vt = (poses.astype(np.float64)[:, None, :] + np.einsum("wij,cj->wci", R, rt.astype(np.float64))).reshape(-1, 3)
    mv = ModelTraj(torch.from_numpy(pts), torch.from_numpy(vt.astype(np.float32)), torch.from_numpy(vq.astype(np.float32)),
                   torch.from_numpy(K), IW, IH, device=dev, occlusion="hpr")
    mv(vis_wps_dist=0.0)
    np.testing.assert_allclose(m.rewards.detach().cpu().numpy(), mv.rewards.detach().cpu().numpy(), rtol=2e-5, atol=0)
    # occlusion changes something: the unoccluded rig model sees more
    m0 = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH,
                   device=dev, rig=(rq, rt))
    m0(vis_wps_dist=0.0)
    assert m.rewards.mean().item() < m0.rewards.mean().item()


def test_occlusion_pad_bits_follow_the_last_sorted_point(dev):
    """The packed cloud is padded with copies of its last sorted point.  When that point is hidden for a waypoint, its copies
    must be hidden too (an occlusion row's pad bits repeat the bit of position n-1) — otherwise a hidden point's p could
    still win the waypoint's max through a pad.  N is not a multiple of the padding; the last sorted point is forced to be the
    cloud's best-seen point and is marked occluded by hand."""
    from oracle import oracle
    from trajectory_optimization_amd import ops
    n = 5_003
    pts = synth.make_cloud(n, seed=29)
    poses, quats = synth.make_path(3, optical=True, jitter_seed=29)
    cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    perm = cloud.perm.cpu().numpy()
    last = int(perm[n - 1])                                   # caller's index of the last sorted point
    # move that point to where waypoint 0 sees best: it becomes the argmax of waypoint 0
    p0 = oracle.pose_forward(pts, poses[0], quats[0], K, IW, IH, prec="f64")[0]
    pts2 = pts.copy()
    pts2[last] = pts[int(np.argmax(p0))] + np.float32(1e-3)
    cloud = ops.PackedCloud(torch.from_numpy(pts2).to(dev), sort=False)   # caller's order kept: `last` stays last only if it is
    # unsorted packing keeps the caller's order, so put the special point at the end
    order = np.r_[np.delete(np.arange(n), last), last]
    pts3 = pts2[order]
    cloud = ops.PackedCloud(torch.from_numpy(pts3).to(dev), sort=False)
    # occlusion rows through the library's own builder: kept = everything, visible = everything but the last point (waypoint 0)
    npad = cloud.npad
    rows = torch.empty((3, npad // 32), dtype=torch.int32, device=dev)
    kept = torch.arange(n, dtype=torch.int32, device=dev).repeat(3, 1).contiguous()
    kcnt = torch.full((3,), n, dtype=torch.int32, device=dev)
    vis_lists = [torch.arange(n - 1, dtype=torch.int32, device=dev), torch.arange(n, dtype=torch.int32, device=dev),
                 torch.arange(n, dtype=torch.int32, device=dev)]
    vis_off = torch.tensor([0, n - 1, 2 * n - 1, 3 * n - 1], dtype=torch.int32, device=dev)
    allv = torch.zeros(3, dtype=torch.int32, device=dev)
    from trajectory_optimization_amd._lib import check, lib, ptr, stream_ptr
    check(lib().tohip_occlusion_rows(n, ptr(cloud.inv_perm), ptr(kept), ptr(kcnt), ptr(torch.cat(vis_lists)), ptr(vis_off), ptr(allv), 3,
                                     ptr(rows), stream_ptr()), "rows")
    bits = ((rows.cpu().numpy().view(np.uint32)[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(3, -1)
    assert bits[0, n - 1] == 0 and not bits[0, n:].any()        # pads of row 0 are hidden like the point they copy
    assert bits[1, n - 1] == 1 and bits[1, n:].all()
    ws = ops.TrajWorkspace(cloud, 3)
    for flags in (0, ops.DENSE):
        lo, mm = ops.traj_forward(cloud, p, q, cam, ws, flags=flags, occ=rows)
        rew, sc = ops.traj_reward(cloud, lo, cam, ws)
        occ = np.ones((3, n), np.float32)
        occ[0, n - 1] = 0.0
        f = oracle.traj_forward(pts3, poses, quats, K, IW, IH, prec="f64", occ=occ)
        np.testing.assert_allclose(mm.cpu().numpy()[:, 1], f["pmax"] - f["pmin"], rtol=1e-5)   # the hidden point did not set the max
        np.testing.assert_allclose(rew.cpu().numpy(), f["rewards"], rtol=1e-5, atol=0)


def test_xy_yaw_gradient_matches_finite_differences(dev):
    """tools.xy_yaw_gradient: (dL/dx, dL/dy, dL/dyaw) from poses.grad / quats.grad, against central differences of the f64
    oracle's visibility loss under a planar move and a turn about the world z axis."""
    from oracle import oracle
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.tools import xy_yaw_gradient
    pts = synth.make_cloud(40_000, seed=33)
    poses, quats = synth.make_path(5, optical=True, jitter_seed=33)
    m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev)
    m(vis_wps_dist=0.0)
    m.loss["vis"].backward()
    g = xy_yaw_gradient(m.poses.grad, m.quats.data, m.quats.grad).cpu().numpy()
    assert g.shape == (5, 3)

    def loss(P, Q):
        return oracle.traj_forward(pts, P, Q, K, IW, IH, prec="f64")["loss_vis"]
    h = 2e-3
    for wi in (0, 2, 4):
        for ci in (0, 1):
            Pp, Pm = poses.astype(np.float64).copy(), poses.astype(np.float64).copy()
            Pp[wi, ci] += h
            Pm[wi, ci] -= h
            fd = (loss(Pp.astype(np.float32), quats) - loss(Pm.astype(np.float32), quats)) / (2 * h)
            assert abs(g[wi, ci] - fd) <= 0.03 * np.abs(g[:, :2]).max() + 1e-7
        def turned(a):
            r = np.array([np.cos(a / 2), 0, 0, np.sin(a / 2)])
            Q = quats.astype(np.float64).copy()
            Q[wi] = synth.quat_mul(r, Q[wi])
            return Q.astype(np.float32)
        fd = (loss(poses, turned(h)) - loss(poses, turned(-h))) / (2 * h)
        assert abs(g[wi, 2] - fd) <= 0.03 * np.abs(g[:, 2]).max() + 1e-7


def test_occlusion_rows_do_not_depend_on_the_chunking(dev, monkeypatch):
    """occlusion_bits sends the waypoints through the cull stage in chunks sized by a memory budget: same rows whatever the budget."""
    from trajectory_optimization_amd import ops
    pts = synth.make_cloud(30_000, seed=44)
    poses, quats = synth.make_path(7, optical=True, jitter_seed=44)
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P)
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    a = ops.occlusion_bits(cloud, P, p, q, cam, 1.0, 15.0, "hpr")
    monkeypatch.setattr(ops, "OCCLUSION_CULL_BYTES", 16 * 30_000 * 2)   # two waypoints per chunk
    b = ops.occlusion_bits(cloud, P, p, q, cam, 1.0, 15.0, "hpr")
    assert torch.equal(a, b)
    assert 0 < int((a != -1).sum())   # something is occluded


def test_occlusion_rows_culled_in_packed_order_equal_the_rows_culled_in_the_callers_order(dev):
    """The refresh culls the cloud in its PACKED order (kept indices = packed positions, ascending: bit rows written run by run,
    tohip_occlusion_rows_masked with inv_perm = NULL); handed a copy of the points it cannot take that shortcut and culls in the
    caller's order (scattered bits through inv_perm).  Same rows, bit for bit — with exact duplicates in the cloud too, where the
    hull reports the lowest row of each set of copies in either order."""
    from trajectory_optimization_amd import ops
    pts = synth.make_cloud(60_000, seed=45)
    pts[1000:1400] = pts[50_000:50_400]          # exact copies, far apart in the caller's order
    poses, quats = synth.make_path(9, optical=True, jitter_seed=45)
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P)
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    a = ops.occlusion_bits(cloud, cloud.points, p, q, cam, 1.0, 15.0, "hpr")
    b = ops.occlusion_bits(cloud, P.clone(), p, q, cam, 1.0, 15.0, "hpr")
    assert torch.equal(a, b)
    assert 0 < int((a != -1).sum())


@pytest.mark.parametrize("dense,lens", [(False, [9, 17, 5]), (True, [9, 17, 5]), (False, [40, 50, 23]), (True, [33, 1, 64])])
def test_several_trajectories_in_one_pass_equal_separate_calls(dev, dense, lens):
    """tohip_traj_*_multi: B trajectories' waypoints as one batch of virtual waypoints, each with its own log-odds vector,
    rewards, loss scalars and gradients — bit for bit what B separate calls give (unequal lengths, a rig, both modes)."""
    from trajectory_optimization_amd import ops
    pts = synth.make_cloud(150_000, seed=61)
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P)
    cam = ops.Camera(K, IW, IH)
    flags = ops.DENSE if dense else 0
    rg = ops.CameraRig(*synth.camera_rig(2), dev)
    # (with two cameras each: the longer sets put trajectories across the 64-waypoint words of the flag rows)
    paths = [synth.make_path(w, optical=True, jitter_seed=70 + i) for i, w in enumerate(lens)]
    # move the second and third path so that the three see different parts of the cloud (and share some slots)
    paths[1] = (paths[1][0] + np.float32([0.0, 4.0, 0.0]), paths[1][1])
    paths[2] = (paths[2][0] * np.float32(0.5), paths[2][1])
    p_all = torch.from_numpy(np.concatenate([p for p, _ in paths])).to(dev)
    q_all = torch.from_numpy(np.concatenate([q for _, q in paths])).to(dev)
    toff = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int32, device=dev)
    B, W = len(lens), sum(lens)
    ws = ops.TrajWorkspace(cloud, W * 2, B)
    half = torch.empty((B, cloud.n), device=dev)
    lo, mm = ops.traj_forward_multi(cloud, p_all, q_all, toff, cam, ws, rig=rg, flags=flags, rewards_half=half)
    rew, sc = ops.traj_reward_multi(cloud, lo, cam, ws, rewards=half, prefilled=True)
    gout = torch.tensor([1.0, 0.5, 2.0], device=dev)
    pg, qg = ops.traj_backward_multi(cloud, W, B, cam, ws, lo, scalars=sc, gout=gout, rig=rg, flags=flags)
    g = torch.rand((B, cloud.n), generator=torch.Generator().manual_seed(3)).to(dev) - 0.3
    pg2, qg2 = ops.traj_backward_multi(cloud, W, B, cam, ws, lo, grad_rewards=g, rig=rg, flags=flags)
    assert float(rew.max()) > 0.5
    o = 0
    for b, w in enumerate(lens):
        wsb = ops.TrajWorkspace(cloud, w * 2)
        p, q = p_all[o:o + w].contiguous(), q_all[o:o + w].contiguous()
        lo1, mm1 = ops.traj_forward(cloud, p, q, cam, wsb, rg, flags=flags)
        rew1, sc1 = ops.traj_reward(cloud, lo1, cam, wsb)
        pg1, qg1 = ops.traj_backward(cloud, w, cam, wsb, lo1, scalars=sc1, gout=gout[b:b + 1].contiguous(), rig=rg, flags=flags)
        pg3, qg3 = ops.traj_backward(cloud, w, cam, wsb, lo1, grad_rewards=g[b].contiguous(), rig=rg, flags=flags)
        assert torch.equal(lo[b], lo1) and torch.equal(mm[2 * o:2 * (o + w)], mm1)
        assert torch.equal(rew[b], rew1) and torch.equal(sc[b], sc1)
        assert torch.equal(pg[o:o + w], pg1) and torch.equal(qg[o:o + w], qg1)
        assert torch.equal(pg2[o:o + w], pg3) and torch.equal(qg2[o:o + w], qg3)
        o += w


def test_optimize_trajectories_equals_independent_runs(dev):
    """optimizer.optimize_trajectories: several candidate trajectories over one cloud, one set of launches per step — every
    model ends up where its own optimize_trajectory run puts it, bit for bit, early stops included."""
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectories, optimize_trajectory
    pts = synth.make_cloud(120_000, seed=62)
    P = torch.from_numpy(pts)
    paths = []
    for i in range(4):
        p, q = synth.make_path(14, optical=True, jitter_seed=80 + i)
        paths.append((p + np.float32([0.0, 1.5 * i - 2.0, 0.0]), q))

    def models():
        return [ModelTraj(P, torch.from_numpy(p), torch.from_numpy(q), torch.from_numpy(K), IW, IH, device=dev) for p, q in paths]
    kw = dict(n_opt_steps=6, lr_pose=0.05, lr_quat=0.01, rewards_th=1.004, smoothness_th=0.5, vis_wps_dist=0.5)
    batch = models()
    res_b = optimize_trajectories(batch, **kw)
    single = models()
    res_s = [optimize_trajectory(m, **kw) for m in single]
    assert any(r.stopped for r in res_s) and not all(r.stopped and r.steps_taken == 1 for r in res_s)
    for mb, ms, rb, rs in zip(batch, single, res_b, res_s):
        assert torch.equal(mb.poses.data, ms.poses.data) and torch.equal(mb.quats.data, ms.quats.data)
        assert torch.equal(mb.rewards, ms.rewards)
        assert rb.steps_taken == rs.steps_taken and rb.stopped == rs.stopped and rb.losses == rs.losses
        for kk in ("vis", "l2", "length", "smooth"):
            assert float(mb.loss[kk]) == float(ms.loss[kk])
    with pytest.raises(ValueError):
        optimize_trajectories([batch[0], ModelTraj(P, torch.from_numpy(paths[0][0][:9]), torch.from_numpy(paths[0][1][:9]),
                                                    torch.from_numpy(K), IW, IH, device=dev)], **kw)


def test_models_share_one_packed_cloud(dev):
    """One packed cloud for many models (the reference builds a model per message over the same map,
    /root/reference/src/trajectory_optimization.py:129-136): ModelTraj(cloud, ...), ModelTraj(points, ..., cloud=model) and
    ModelTraj.sharing_cloud_of pack nothing, hold the SAME blob, and give the bits of a model that packed its own copy."""
    from trajectory_optimization_amd import ops
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectories
    pts = synth.make_cloud(60_000, seed=64)
    P = torch.from_numpy(pts).to(dev)
    paths = [synth.make_path(11, optical=True, jitter_seed=90 + i) for i in range(3)]
    Kt = torch.from_numpy(K)
    own = [ModelTraj(P, torch.from_numpy(p), torch.from_numpy(q), Kt, IW, IH, device=dev) for p, q in paths]
    cloud = ops.PackedCloud(P)
    m0 = ModelTraj(cloud, torch.from_numpy(paths[0][0]), torch.from_numpy(paths[0][1]), Kt, IW, IH, device=dev)
    m1 = ModelTraj(P, torch.from_numpy(paths[1][0]), torch.from_numpy(paths[1][1]), Kt, IW, IH, device=dev, cloud=m0)
    m2 = ModelTraj.sharing_cloud_of(m0, torch.from_numpy(paths[2][0]), torch.from_numpy(paths[2][1]))
    shared = [m0, m1, m2]
    assert all(m._cloud is cloud for m in shared) and all(m.points.data_ptr() == P.data_ptr() for m in shared)
    for a, b in zip(own, shared):
        la, lb = a(vis_wps_dist=0.0), b(vis_wps_dist=0.0)
        la.backward()
        lb.backward()
        assert torch.equal(la, lb) and torch.equal(a.rewards, b.rewards)
        assert torch.equal(a.poses.grad, b.poses.grad) and torch.equal(a.quats.grad, b.quats.grad)
    kw = dict(n_opt_steps=4, lr_pose=0.05, lr_quat=0.01, rewards_th=1e9, vis_wps_dist=0.0)
    ra, rb = optimize_trajectories(own, **kw), optimize_trajectories(shared, **kw)
    for a, b, x, y in zip(own, shared, ra, rb):
        assert torch.equal(a.poses.data, b.poses.data) and x.losses == y.losses
    with pytest.raises(ValueError):
        ModelTraj(ops.PackedCloud(P, sort=False), torch.from_numpy(paths[0][0]), torch.from_numpy(paths[0][1]), Kt, IW, IH, device=dev)
    with pytest.raises(ValueError):
        ModelTraj(P[:100], torch.from_numpy(paths[0][0]), torch.from_numpy(paths[0][1]), Kt, IW, IH, device=dev, cloud=cloud)
    with pytest.raises(ValueError):   # an equal-sized OTHER cloud must not be replaced silently by the packed one
        ModelTraj(P + 0.25, torch.from_numpy(paths[0][0]), torch.from_numpy(paths[0][1]), Kt, IW, IH, device=dev, cloud=cloud)
    # the same rows in another tensor (e.g. still on the host) are fine
    ModelTraj(P.cpu().clone(), torch.from_numpy(paths[0][0]), torch.from_numpy(paths[0][1]), Kt, IW, IH, device=dev, cloud=cloud)


def test_outputs_left_to_the_last_step_change_nothing(dev):
    """TOHIP_TRAJ_OPT_LAST_OUTPUTS (optimize_trajectory's default): the N-sized outputs — rewards, log-odds — are written by the run's
    last step only; every step computes them.  Poses, losses, stop step and the rewards handed back are those of a run whose
    every step writes everything, bit for bit; a run cut short (run(n) with n below the planned steps) writes them every step."""
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd import optimizer
    pts = torch.from_numpy(synth.make_cloud(90_000, seed=65)).to(dev)
    paths = [synth.make_path(12, optical=True, jitter_seed=95 + i) for i in range(3)]
    Kt = torch.from_numpy(K)

    def models():
        m0 = ModelTraj(pts, torch.from_numpy(paths[0][0]), torch.from_numpy(paths[0][1]), Kt, IW, IH, device=dev)
        return [m0] + [ModelTraj.sharing_cloud_of(m0, torch.from_numpy(p), torch.from_numpy(q)) for p, q in paths[1:]]
    args = (7, 0.05, 0.01, 1.003, 0.5, 0.0, (0.9, 0.999), 1e-8)
    for B in (1, 3):
        lean, full = optimizer._OptRun(models()[:B], *args), optimizer._OptRun(models()[:B], *args)
        assert lean.c.flags & optimizer.LAST_OUTPUTS
        full.c.flags &= ~optimizer.LAST_OUTPUTS
        lean.rewards.fill_(-1.0)
        lean.run(7)
        full.run(7)
        ra, rb = lean.results(7), full.results(7)
        assert torch.equal(lean.rewards, full.rewards) and torch.equal(lean.lo_sum, full.lo_sum) and float(lean.rewards.min()) >= 0.5
        for a, b, x, y in zip(lean.models, full.models, ra, rb):
            assert torch.equal(a.poses.data, b.poses.data) and torch.equal(a.quats.data, b.quats.data)
            assert x.losses == y.losses and x.steps_taken == y.steps_taken and x.stopped == y.stopped
        short = optimizer._OptRun(models()[:B], *args)
        short.run(4)   # not the planned seven: the flag is dropped, the rewards are there
        ref = optimizer._OptRun(models()[:B], 4, *args[1:])
        ref.run(4)
        assert torch.equal(short.rewards, ref.rewards)


@pytest.mark.parametrize("method", ["zbuffer", "hpr"])
def test_occlusion_motion_triggered_refresh(dev, method):
    """ModelTraj(occlusion_refresh_tol=...): rows are rebuilt for the waypoints that have moved by more than the tolerance since
    THEIR rows were built — nothing while nobody has, only the movers' rows when somebody has (bit for bit the rows a fresh model
    builds at the new poses), everything when occlusion_refresh_every caps the age."""
    from trajectory_optimization_amd.model import ModelTraj
    pts = torch.from_numpy(synth.make_cloud(60_000, seed=34))
    poses, quats = synth.make_path(5, optical=True, jitter_seed=34)

    def model(p, **kw):
        return ModelTraj(pts, torch.from_numpy(p), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev, occlusion=method, **kw)
    m = model(poses, occlusion_refresh_every=100, occlusion_refresh_tol=0.05, occlusion_check_every=1)
    with torch.no_grad():
        m(vis_wps_dist=0.0)
        rows0 = m._occ_cache[0]
        assert m.occlusion_rebuilds == [1, 0]
        m(vis_wps_dist=0.0)                                    # nobody has moved
        assert m._occ_cache[0] is rows0 and m.occlusion_rebuilds == [1, 0]
        m.poses.data[1, 0] += 0.01                             # below the tolerance
        m(vis_wps_dist=0.0)
        assert m._occ_cache[0] is rows0 and m.occlusion_rebuilds == [1, 0]
        m.poses.data[3, 1] += 0.3                              # one waypoint beyond it
        loss = m(vis_wps_dist=0.0)
        assert m.occlusion_rebuilds == [1, 1] and m._occ_cache[0] is not rows0
        moved = poses.copy()
        moved[3, 1] += 0.3
        fresh = model(moved)
        fresh(vis_wps_dist=0.0)
        assert torch.equal(m._occ_cache[0][3], fresh._occ_cache[0][3])           # the mover's row: what a fresh model builds there
        keep = [0, 1, 2, 4]
        assert torch.equal(m._occ_cache[0][keep], rows0[keep])                    # the others: kept
        assert bool((m._occ_cache[0][3] != rows0[3]).any())
        m.quats.data[0] = torch.nn.functional.normalize(m.quats.data[0] + torch.tensor([0.0, 0.1, 0.0, 0.0], device=dev), dim=0)   # a turn of ~0.2 rad
        m(vis_wps_dist=0.0)
        assert m.occlusion_rebuilds == [1, 2]
        cap = model(poses, occlusion_refresh_every=2, occlusion_refresh_tol=1e9)
        for _ in range(3):
            cap(vis_wps_dist=0.0)
        assert cap.occlusion_rebuilds == [2, 0]
        assert torch.isfinite(loss)
