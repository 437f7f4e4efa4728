#!/usr/bin/env python3
"""Generate golden vectors from the reference itself (this container only).

Runs /root/reference/src/{model,tools}.py UNMODIFIED, on CPU, behind stubs for
the ROS modules it imports but never uses on the hot path, and stores inputs +
outputs as small .npz fixtures next to this script.  The reference never
travels: only these data files are committed.

Third-party arithmetic that is not installed here and therefore restated:
pytorch3d.transforms.{quaternion_invert, quaternion_apply} (pytorch3d 0.3.0,
/root/reference/requirements.txt:2; call sites /root/reference/src/model.py:4,54,56
and /root/reference/src/pc_processor.py:9,68-69).  The shim below follows the
published pytorch3d definition (real-first Hamilton product evaluated left to
right in the tensors' dtype); parity at that boundary is pinned by nothing in the
reference's own tests.

Usage:  python tests/golden/make_golden.py      (needs /root/reference)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------- stubs
class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        return _Stub(self.__name__ + "()")


for _name in ["rospy", "cv_bridge", "tf2_ros", "tf", "nav_msgs", "nav_msgs.msg", "sensor_msgs",
              "sensor_msgs.msg", "geometry_msgs", "geometry_msgs.msg", "std_msgs", "std_msgs.msg"]:
    sys.modules[_name] = _Stub(_name)


class _PointField:
    """sensor_msgs/PointField: the message definition's datatype constants (what pointcloud_utils.py keys its
    numpy dtype tables on) + the four data members."""
    INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 = 1, 2, 3, 4, 5, 6, 7, 8

    def __init__(self, name="", offset=0, datatype=0, count=1):
        self.name, self.offset, self.datatype, self.count = name, offset, datatype, count


class _PointCloud2(types.SimpleNamespace):
    pass


sys.modules["sensor_msgs.msg"].PointField = _PointField
sys.modules["sensor_msgs.msg"].PointCloud2 = _PointCloud2


def _quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)


def _quaternion_invert(q):
    return q * q.new_tensor([1, -1, -1, -1])


def _quaternion_apply(q, point):
    real = point.new_zeros(point.shape[:-1] + (1,))
    pq = torch.cat((real, point), -1)
    out = _quaternion_raw_multiply(_quaternion_raw_multiply(q, pq), _quaternion_invert(q))
    return out[..., 1:]


_p3d = types.ModuleType("pytorch3d")
_p3d_t = types.ModuleType("pytorch3d.transforms")
_p3d_t.quaternion_invert = _quaternion_invert
_p3d_t.quaternion_apply = _quaternion_apply
_p3d.transforms = _p3d_t
sys.modules["pytorch3d"] = _p3d
sys.modules["pytorch3d.transforms"] = _p3d_t

np.float = float  # /root/reference/src/pointcloud_utils.py:180,200,220 default args
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))

import tools as ref_tools  # noqa: E402
import model as ref_model  # noqa: E402

from trajectory_optimization_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
CPU = torch.device("cpu")
K, IMG_W, IMG_H = ref_tools.load_intrinsics(CPU)
assert np.array_equal(K.numpy(), synth.K_INTRINS) and IMG_W == synth.IMG_WIDTH and IMG_H == synth.IMG_HEIGHT


_BUNDLED = {}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if k == "points" and _BUNDLED and v is _BUNDLED["pts"]:
            out["points_ref"] = np.asarray("bundled")  # stored once in bundled.npz
            continue
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  keys={sorted(out)}")


def bundled():
    """The reference's one sample cloud/path pair (data/points/point_cloud_10.npz,
    data/paths/path_poses_10.npz), cast to f32 as its drivers do; stored once."""
    if not _BUNDLED:
        pts = np.load(os.path.join(REF, "data/points/point_cloud_10.npz"))["pts"].astype(np.float32)
        poses = np.load(os.path.join(REF, "data/paths/path_poses_10.npz"))["poses"].astype(np.float32)
        save("bundled", pts=pts, poses=poses)
        _BUNDLED.update(pts=pts, poses=poses)
    return _BUNDLED["pts"], _BUNDLED["poses"]


def run_traj(pts, poses, quats, vis_wps_dist, **kw):
    m = ref_model.ModelTraj(points=torch.from_numpy(pts), wps_poses=torch.from_numpy(poses),
                            wps_quats=torch.from_numpy(quats), intrins=K, img_width=IMG_W,
                            img_height=IMG_H, device=CPU, **kw)
    loss = m(vis_wps_dist=vis_wps_dist)
    loss.backward()
    mean_d = (m.poses0[1:] - m.poses0[:-1]).norm(dim=1).mean() if len(poses) > 1 else torch.tensor(float("nan"))
    return dict(points=pts, poses=poses, quats=quats, vis_wps_dist=np.float64(vis_wps_dist),
                loss=loss, loss_vis=m.loss["vis"], loss_l2=m.loss["l2"], loss_smooth=m.loss["smooth"],
                loss_length=m.loss["length"], rewards=m.rewards, poses_grad=m.poses.grad,
                quats_grad=m.quats.grad, mean_wps_dist=mean_d,
                **{k: np.float64(v) for k, v in kw.items()})


def run_traj_visonly(pts, poses, quats):
    """Visibility term alone (vis = 1/(mean(rewards)+eps)): gradient of the HIP
    path proper, without the O(W) regularisers."""
    m = ref_model.ModelTraj(points=torch.from_numpy(pts), wps_poses=torch.from_numpy(poses),
                            wps_quats=torch.from_numpy(quats), intrins=K, img_width=IMG_W,
                            img_height=IMG_H, device=CPU)
    m(vis_wps_dist=0.0)
    m.loss["vis"].backward()
    return dict(vis_poses_grad=m.poses.grad, vis_quats_grad=m.quats.grad)


def gen_traj():
    pts, poses = bundled()
    quats = np.tile(np.array([[1, 0, 0, 0]], dtype=np.float32), (len(poses), 1))
    save("traj_bundled_default", **run_traj(pts, poses, quats, 0.5))
    # every waypoint evaluated, tilted non-unit quaternions
    rng = np.random.default_rng(7)
    q2 = (quats + 0.3 * rng.standard_normal(quats.shape)).astype(np.float32) * 1.7
    d = run_traj(pts, poses, q2, 0.0, smoothness_weight=28.0, traj_length_weight=0.05)
    d.update(run_traj_visonly(pts, poses, q2))
    save("traj_bundled_tilted_all", **d)
    for n, w, seed in [(1000, 1, 11), (1000, 3, 12), (10000, 8, 13), (20000, 32, 14)]:
        cloud = synth.make_cloud(n, seed=seed)
        p, q = synth.make_path(w, optical=True, jitter_seed=seed)
        if w < 3:
            # criterion needs >= 3 waypoints for mean_angle_calc; use vis term only
            m = ref_model.ModelTraj(points=torch.from_numpy(cloud), wps_poses=torch.from_numpy(p),
                                    wps_quats=torch.from_numpy(q), intrins=K, img_width=IMG_W,
                                    img_height=IMG_H, device=CPU)
            # forward() calls criterion which divides by (N_wps-2)=-1 for W=1: still finite
            try:
                loss = m(vis_wps_dist=0.0)
            except Exception as e:  # pragma: no cover
                print("W<3 forward raised", e)
                continue
            m.zero_grad()
            m.loss["vis"].backward()
            save(f"traj_synth_{n}x{w}", points=cloud, poses=p, quats=q, vis_wps_dist=np.float64(0.0),
                 loss=loss, loss_vis=m.loss["vis"], rewards=m.rewards,
                 vis_poses_grad=m.poses.grad, vis_quats_grad=m.quats.grad)
            continue
        d = run_traj(cloud, p, q, 0.0)
        d.update(run_traj_visonly(cloud, p, q))
        save(f"traj_synth_{n}x{w}", **d)
    # duplicated points -> ties at the per-waypoint max (torch shares the gradient evenly)
    cloud = synth.make_cloud(2000, seed=21)
    cloud = np.concatenate([cloud, cloud[:500]], axis=0)
    p, q = synth.make_path(4, optical=True, jitter_seed=21)
    d = run_traj(cloud, p, q, 0.0)
    d.update(run_traj_visonly(cloud, p, q))
    save("traj_synth_ties", **d)
    # compact cloud right in front of the camera: min(p) > 0, many active points
    rng = np.random.default_rng(22)
    cloud = (rng.random((3000, 3)) * np.array([2.0, 2.0, 1.0]) + np.array([2.0, -1.0, -0.5])).astype(np.float32)
    p = np.array([[0, 0, 0], [0.3, 0.1, 0], [0.6, -0.1, 0.05]], dtype=np.float32)
    q = np.tile(synth.Q_OPTICAL.astype(np.float32)[None], (3, 1))
    d = run_traj(cloud, p, q, 0.0)
    d.update(run_traj_visonly(cloud, p, q))
    save("traj_synth_dense", **d)


def gen_ties():
    """Three- and five-fold copies of a block of points: every waypoint's argmax (and the other extremal / active points of
    the block) is a tie set of >= 3 elements — what torch.max / torch.min do with the gradient there (model.py:226-227)."""
    cloud = synth.make_cloud(1500, seed=23)
    cloud = np.concatenate([cloud, cloud[:400], cloud[:400], cloud[100:200], cloud[100:200]], axis=0)   # 3x [0,400), 5x [100,200)
    p, q = synth.make_path(5, optical=True, jitter_seed=23)
    d = run_traj(cloud, p, q, 0.0)
    d.update(run_traj_visonly(cloud, p, q))
    save("traj_synth_ties3", **d)


def gen_clip():
    """Non-default pc_clip_limits (min_dist, max_dist): the Gaussian of get_dist_mask moves to (0.5+8)/2 = 4.25 with
    std 3.75 (model.py:18-21).  python tests/golden/make_golden.py clip"""
    cloud = synth.make_cloud(6000, seed=31)
    p, q = synth.make_path(5, optical=True, jitter_seed=31)
    d = run_traj(cloud, p, q, 0.0, min_dist=0.5, max_dist=8.0)

    def visonly():
        m = ref_model.ModelTraj(points=torch.from_numpy(cloud), wps_poses=torch.from_numpy(p), wps_quats=torch.from_numpy(q),
                                intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU, min_dist=0.5, max_dist=8.0)
        m(vis_wps_dist=0.0)
        m.loss["vis"].backward()
        return dict(vis_poses_grad=m.poses.grad, vis_quats_grad=m.quats.grad)
    d.update(visonly())
    save("traj_synth_clip", **d)
    mp = ref_model.ModelPose(points=torch.from_numpy(cloud), trans0=torch.tensor([[1.0, -2.0, 0.3]]),
                             q0=torch.from_numpy(q[2:3].copy()), intrins=K, img_width=IMG_W, img_height=IMG_H,
                             min_dist=0.5, max_dist=8.0, device=CPU)
    loss = mp()
    loss.backward()
    save("pose_synth_clip", points=cloud, trans0=np.array([[1.0, -2.0, 0.3]], np.float32), q0=q[2:3].copy(), hpr=np.asarray(False),
         min_dist=np.float64(0.5), max_dist=np.float64(8.0), loss=loss, observations=mp.observations, trans_grad=mp.trans.grad,
         quat_grad=mp.quat.grad)


def gen_adam():
    """Parameter values after k Adam steps on the bundled cloud/path
    (/root/reference/src/trajectory_optimization.py:91-124 loop, launch-file rates)."""
    pts, poses = bundled()
    quats = np.tile(np.array([[1, 0, 0, 0]], dtype=np.float32), (len(poses), 1))
    m = ref_model.ModelTraj(points=torch.from_numpy(pts), wps_poses=torch.from_numpy(poses),
                            wps_quats=torch.from_numpy(quats), intrins=K, img_width=IMG_W,
                            img_height=IMG_H, device=CPU)
    opt = torch.optim.Adam([{"params": [m.poses], "lr": 0.12}, {"params": [m.quats], "lr": 0.05}])
    out = dict(points=pts, poses=poses, quats=quats, lr_pose=0.12, lr_quat=0.05)
    losses, mean_rewards = [], []
    for i in range(10):
        opt.zero_grad()
        loss = m()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        mean_rewards.append(m.rewards.mean().item())
        if i + 1 in (1, 5, 10):
            out[f"poses_step{i + 1}"] = m.poses.detach().clone()
            out[f"quats_step{i + 1}"] = m.quats.detach().clone()
    out["losses"] = np.array(losses)
    out["mean_rewards"] = np.array(mean_rewards)
    save("traj_adam_bundled", **out)

    # pose optimisation loop (/root/reference/src/pose_optimization.py:82-136, launch rates 0.02/0.02)
    t0 = np.array([[6.0, 2.0, 0.0]], dtype=np.float32)
    q0 = np.array([[1.0, 0.0, 0.0, 0.0]], dtype=np.float32)
    # clones: on CPU the reference's Parameters alias the tensors it is given (model.py:86-89), so Adam would
    # otherwise overwrite the t0/q0 arrays saved below
    mp = ref_model.ModelPose(points=torch.from_numpy(pts), trans0=torch.from_numpy(t0.copy()), q0=torch.from_numpy(q0.copy()),
                             intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU)
    opt = torch.optim.Adam([{"params": [mp.trans], "lr": 0.02}, {"params": [mp.quat], "lr": 0.02}])
    out = dict(points=pts, trans0=t0, q0=q0, lr_pose=0.02, lr_quat=0.02)
    losses = []
    for i in range(10):
        opt.zero_grad()
        loss = mp()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if i + 1 in (1, 5, 10):
            out[f"trans_step{i + 1}"] = mp.trans.detach().clone()
            out[f"quat_step{i + 1}"] = mp.quat.detach().clone()
    out["losses"] = np.array(losses)
    save("pose_adam_bundled", **out)


def gen_pose():
    pts, _ = bundled()
    for tag, t0, q0, hpr in [("nohpr", [6.0, 2.0, 0.0], [1.0, 0.0, 0.0, 0.0], False),
                             ("hpr", [6.0, 2.0, 0.0], [1.0, 0.0, 0.0, 0.0], True),
                             ("tilted", [10.0, 12.0, 0.5], [0.9, 0.1, -0.5, 0.3], False)]:
        t0a = np.array([t0], dtype=np.float32)
        q0a = np.array([q0], dtype=np.float32)
        m = ref_model.ModelPose(points=torch.from_numpy(pts), trans0=torch.from_numpy(t0a),
                                q0=torch.from_numpy(q0a), intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU)
        loss = m(hpr=hpr)
        loss.backward()
        save(f"pose_bundled_{tag}", points=pts, trans0=t0a, q0=q0a, hpr=np.bool_(hpr), loss=loss,
             observations=m.observations, trans_grad=m.trans.grad, quat_grad=m.quat.grad)
    # BASELINE.json config 1: 10k synthetic cloud, one pose, HPR + reward
    cloud = synth.make_cloud(10000, seed=0)
    p, q = synth.make_path(1, optical=True)
    m = ref_model.ModelPose(points=torch.from_numpy(cloud), trans0=torch.from_numpy(p), q0=torch.from_numpy(q),
                            intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU)
    loss = m(hpr=True)
    loss.backward()
    save("pose_synth_10k_hpr", points=cloud, trans0=p, q0=q, hpr=np.bool_(True), loss=loss,
         observations=m.observations, trans_grad=m.trans.grad, quat_grad=m.quat.grad)


def gen_nothing():
    """A pose that sees nothing (/root/reference/src/model.py:124-127: loss = 1/(sum + eps) = 1/eps): a cloud 26.8-28.7 m from the
    Gaussian's centre, where every observation is below FLT_MIN — subnormal, or zero.  The reference on a CPU adds the subnormals
    up (sum ~1e-39) and hands back a gradient of ~1e-26; the GPU's exp unit flushes them to zero.  What is pinned: the loss, and
    the SIZE of the gradient (the HIP path's bound: loss equal, |gradient| <= 1e-20)."""
    rng = np.random.default_rng(11)
    cloud = (np.float32([30.7, 3.0, 3.0]) + rng.uniform(-0.9, 0.9, (2000, 3)).astype(np.float32)).astype(np.float32)
    t0a, q0a = np.zeros((1, 3), np.float32), np.float32([[1.0, 0.0, 0.0, 0.0]])
    m = ref_model.ModelPose(points=torch.from_numpy(cloud), trans0=torch.from_numpy(t0a), q0=torch.from_numpy(q0a), intrins=K,
                            img_width=IMG_W, img_height=IMG_H, device=CPU)
    loss = m()
    loss.backward()
    obs = m.observations.detach().numpy()
    tiny = np.finfo(np.float32).tiny
    assert obs.max() < tiny and (obs > 0).any(), (obs.max(), (obs > 0).sum())
    save("pose_sees_nothing", points=cloud, trans0=t0a, q0=q0a, hpr=np.bool_(False), loss=loss, observations=m.observations,
         trans_grad=m.trans.grad, quat_grad=m.quat.grad, n_subnormal=np.int64((obs > 0).sum()), obs_sum=np.float64(obs.astype(np.float64).sum()))
    print("pose_sees_nothing: loss", float(loss), "sum of observations", float(obs.astype(np.float64).sum()), "subnormal observations", int((obs > 0).sum()),
          "| trans grad", m.trans.grad.numpy(), "quat grad", m.quat.grad.numpy())


def gen_funcs():
    cloud = synth.make_cloud(4096, seed=3)
    p, q = synth.make_path(3, optical=True, jitter_seed=3)
    pts = torch.from_numpy(cloud)
    cam = ref_model.to_camera_frame(pts, torch.from_numpy(q[1:2]), torch.from_numpy(p[1:2]))
    dist = ref_model.get_dist_mask(cam, 1.0, 5.0)
    fov = ref_model.get_fov_mask(cam, IMG_H, IMG_W, K, eps=1e-6)
    fov_bin = ref_model.get_fov_mask(cam, IMG_H, IMG_W, K, binary=True)
    flipped = ref_tools.sphericalFlip(pts, CPU, 2)
    traj = torch.from_numpy(p)
    path27 = torch.from_numpy(bundled()[1])
    save("funcs", points=cloud, quat=q[1:2], trans=p[1:2], cam=cam, dist_mask=dist, fov_mask=fov,
         fov_mask_binary=fov_bin, flipped=flipped, traj=p, traj_length=ref_model.length_calc(traj),
         traj_mean_angle=ref_model.mean_angle_calc(traj), path27=path27,
         path27_length=ref_model.length_calc(path27), path27_mean_angle=ref_model.mean_angle_calc(path27))


def ego_to_cam(points, trans, quat):
    """/root/reference/src/pc_processor.py:63-70 (staticmethod body; the class
    itself needs a live ROS master to construct)."""
    points = points - trans
    quat_inv = _quaternion_invert(quat)
    points = _quaternion_apply(quat_inv, points)
    return points.T.float()


def gen_hard():
    pts, _ = bundled()
    P = torch.from_numpy(pts)
    # row J on the world-frame bundled cloud (what ModelPose(hpr=True) does)
    vis, mask = ref_tools.hidden_pts_removal(P, CPU)
    save("hpr_bundled_world", points=pts, visible_idx=np.flatnonzero(mask.numpy()).astype(np.int32),
         n_visible=np.int64(vis.shape[0]))
    # row L: hard pipeline on the bundled cloud
    q = torch.tensor([0.7071068, 0.0, -0.7071068, 0.0])
    t = torch.tensor([[14.0, 12.0, 0.0]])
    cam = ego_to_cam(P, t, q)  # 3 x N
    kept, dmask, fmask = ref_tools.get_cam_frustum_pts(cam.clone(), IMG_H, IMG_W, K, 1.0, 10.0)
    both = torch.logical_and(dmask, fmask)
    vis2, mask2 = ref_tools.hidden_pts_removal(kept, CPU)
    save("hard_pipeline_bundled", points=pts, quat=q, trans=t, cam=cam, min_dist=1.0, max_dist=10.0,
         dist_mask=np.packbits(dmask.numpy()), fov_mask=np.packbits(fmask.numpy()),
         kept_idx=np.flatnonzero(both.numpy()).astype(np.int32), kept_pts=kept,
         hpr_visible_idx=np.flatnonzero(mask2.numpy()).astype(np.int32), hpr_visible_pts=vis2)
    # synthetic clouds, several viewpoints, incl. a cloud surrounding the viewpoint (Q3: origin interior)
    for name, n, seed, centre in [("hpr_synth_10k", 10000, 0, (0.0, 0.0, 0.0)),
                                  ("hpr_synth_100k", 100000, 5, (0.0, 0.0, 0.0)),
                                  ("hpr_synth_outside", 20000, 6, (30.0, 5.0, 1.0))]:
        cloud = synth.make_cloud(n, seed=seed) - np.asarray(centre, dtype=np.float32)
        C = torch.from_numpy(cloud)
        flipped = ref_tools.sphericalFlip(C, CPU, 2)
        hull = ref_tools.convexHull(flipped, CPU)
        vis, mask = ref_tools.hidden_pts_removal(C, CPU)
        save(name, seed=seed, n=n, centre=np.asarray(centre, dtype=np.float32),
             visible_idx=np.flatnonzero(mask.numpy()).astype(np.int32),
             hull_vertices=hull.vertices.astype(np.int32),
             origin_is_vertex=np.bool_(hull.vertices[-1] == n),
             flipped_head=flipped[:64])
    # exact duplicate rows (raw lidar clouds have them), below and above the size at which the GPU build starts on a sample
    for name, n, seed in [("hpr_synth_dups_20k", 20000, 11), ("hpr_synth_dups_120k", 120000, 12)]:
        centre = np.asarray((30.0, 5.0, 1.0), dtype=np.float32)
        base = synth.make_cloud(n, seed=seed) - centre
        _, m0 = ref_tools.hidden_pts_removal(torch.from_numpy(base), CPU)
        vis0 = np.flatnonzero(m0.numpy())
        rng = np.random.default_rng(seed + 500)
        dup = np.concatenate([vis0[rng.integers(0, len(vis0), 300)], rng.integers(0, n, 300)])   # copies of visible points and of others
        pts, src = synth.with_duplicate_rows(base, dup, seed + 1000)
        C = torch.from_numpy(pts)
        hull = ref_tools.convexHull(ref_tools.sphericalFlip(C, CPU, 2), CPU)
        vis, mask = ref_tools.hidden_pts_removal(C, CPU)
        save(name, seed=seed, n=n, centre=centre, dup=dup.astype(np.int32), n_rows=np.int64(len(pts)),
             visible_idx=np.flatnonzero(mask.numpy()).astype(np.int32), hull_vertices=hull.vertices.astype(np.int32),
             origin_is_vertex=np.bool_(hull.vertices[-1] == len(pts)))
    # origin strictly interior: a shell of points around the viewpoint
    rng = np.random.default_rng(9)
    d = rng.standard_normal((5000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    shell = (d * (3.0 + 2.0 * rng.random((5000, 1)))).astype(np.float32)
    S = torch.from_numpy(shell)
    hull = ref_tools.convexHull(ref_tools.sphericalFlip(S, CPU, 2), CPU)
    vis, mask = ref_tools.hidden_pts_removal(S, CPU)
    save("hpr_shell_origin_inside", points=shell, hull_vertices=hull.vertices.astype(np.int32),
         visible_idx=np.flatnonzero(mask.numpy()).astype(np.int32),
         origin_is_vertex=np.bool_(hull.vertices[-1] == len(shell)))
    # hard frustum on a synthetic camera-frame cloud with the other limits
    cloud = synth.make_cloud(50000, seed=8, extent=(30.0, 30.0, 30.0))
    camf = torch.from_numpy(np.ascontiguousarray(cloud.T))
    for lim in [(1.0, 10.0), (1.0, 15.0)]:
        kept, dmask, fmask = ref_tools.get_cam_frustum_pts(camf.clone(), IMG_H, IMG_W, K, *lim)
        save(f"frustum_synth_{int(lim[1])}", seed=8, n=50000, min_dist=lim[0], max_dist=lim[1],
             dist_mask=np.packbits(dmask.numpy()), fov_mask=np.packbits(fmask.numpy()),
             kept_idx=np.flatnonzero(torch.logical_and(dmask, fmask).numpy()).astype(np.int32))


def gen_ingest():
    import pointcloud_utils as ref_pcu
    rng = np.random.default_rng(17)
    n = 5000
    pts = (rng.standard_normal((n, 3)) * np.array([10.0, 10.0, 2.0])).astype(np.float32)
    pts[rng.integers(0, n, 200), rng.integers(0, 3, 200)] = np.nan
    pts[rng.integers(0, n, 50), rng.integers(0, 3, 50)] = np.inf
    out = {}
    # layout A: x,y,z,intensity f32, 20-byte stride with padding (a typical lidar driver layout)
    recA = np.zeros(n, dtype=np.dtype({"names": ["x", "y", "z", "intensity"], "formats": ["<f4"] * 4,
                                       "offsets": [0, 4, 8, 16], "itemsize": 20}))
    recA["x"], recA["y"], recA["z"], recA["intensity"] = pts[:, 0], pts[:, 1], pts[:, 2], rng.random(n)
    # layout B: float64 coordinates in z,y,x order
    recB = np.zeros(n, dtype=np.dtype({"names": ["z", "y", "x"], "formats": ["<f8"] * 3, "offsets": [0, 8, 16], "itemsize": 24}))
    recB["x"], recB["y"], recB["z"] = pts[:, 0].astype(np.float64) * 1.000000123, pts[:, 1], pts[:, 2]
    for tag, rec, names, dt in (("A", recA, ["x", "y", "z", "intensity"], 7), ("B", recB, ["z", "y", "x"], 8)):
        msg = _PointCloud2(height=1, width=n, point_step=rec.dtype.itemsize, is_bigendian=False, data=rec.tobytes(),
                           fields=[_PointField(nm, rec.dtype.fields[nm][1], dt, 1) for nm in names])
        xyz = ref_pcu.pointcloud2_to_xyz_array(msg)
        out[f"data_{tag}"] = np.frombuffer(rec.tobytes(), np.uint8)
        out[f"point_step_{tag}"] = rec.dtype.itemsize
        out[f"offsets_{tag}"] = np.array([rec.dtype.fields[c][1] for c in "xyz"])
        out[f"datatype_{tag}"] = dt
        out[f"xyz_{tag}"] = xyz  # float64, as the reference returns it
    out["n"] = n
    # the reference's own writer round-trips
    m = ref_pcu.xyz_array_to_pointcloud2(np.nan_to_num(pts[:100], posinf=1.0))
    out["writer_data"] = np.frombuffer(m.data, np.uint8)
    out["writer_point_step"] = m.point_step
    # pc_to_voxel on finite points
    fin = pts[np.isfinite(pts).all(1)] + np.array([20.0, 0.0, 0.0], np.float32)
    vox = ref_pcu.pc_to_voxel(fin, resolution=0.5, x=(0, 40), y=(-20, 20), z=(-4.5, 5.5))
    out["vox_points"] = fin
    out["vox_idx"] = np.argwhere(vox > 0).astype(np.int32)
    out["vox_shape"] = np.array(vox.shape)
    save("ingest", **out)


def gen_dense():
    """The reference's own f32 results where the 1e-5 gradient bar is conditional (tests/test_hip_reference_dense.py): a dense
    room — 200 k points in 6 x 6 x 3 m, 8 waypoints, ~1 850 points per cubic metre — and the two configurations of
    tests/test_hip_conditioning.py's generator in which a waypoint has a point within f32 rounding of p_hat = 1/2
    (/root/reference/src/model.py:229).  The clouds are stored by recipe (seeded generators of trajectory_optimization_amd.synth)
    with a checksum.  python tests/golden/make_golden.py dense"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_hip_conditioning import configurations

    def ref_vis(pts, poses, quats, clip):
        m = ref_model.ModelTraj(points=torch.from_numpy(pts), wps_poses=torch.from_numpy(poses), wps_quats=torch.from_numpy(quats),
                                intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU, min_dist=clip[0], max_dist=clip[1])
        m(vis_wps_dist=0.0)
        m.loss["vis"].backward()
        return dict(loss_vis=m.loss["vis"], rewards=m.rewards, vis_poses_grad=m.poses.grad, vis_quats_grad=m.quats.grad)

    n, w, ext = 200_000, 8, (6.0, 6.0, 3.0)
    pts = synth.make_cloud(n, seed=0, extent=ext)
    poses, quats = synth.make_path(w, optical=True, scale=ext[0] / 40.0)
    save("traj_dense_room_200k", recipe=np.asarray("room"), n=n, seed=0, extent=np.asarray(ext), poses=poses, quats=quats,
         min_dist=np.float64(1.0), max_dist=np.float64(5.0), points_checksum=np.float64(pts.astype(np.float64).sum()),
         **ref_vis(pts, poses, quats, (1.0, 5.0)))
    for it, pts, poses, quats, clip, _dense in configurations():
        if it in (32, 34):
            save(f"traj_conditioning_{it}", recipe=np.asarray("conditioning"), index=it, poses=poses, quats=quats,
                 min_dist=np.float64(clip[0]), max_dist=np.float64(clip[1]), points_checksum=np.float64(pts.astype(np.float64).sum()),
                 **ref_vis(pts, poses, quats, clip))


def gen_stress():
    """The reference's own f32 results on what tools/stress_models.py found outside the bars (test_hip_conditioning.STRESS_CASES):
    amplification just below the upper threshold, cancellation in r (1 - r), and a waypoint whose p underflows to 0 for every point
    (0 / 0).  Clouds by recipe + checksum.  python tests/golden/make_golden.py stress"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_hip_conditioning import STRESS_CASES, stress_case
    for (seed, index), what in STRESS_CASES.items():
        _, pts, poses, quats, clip, _dense, _j = stress_case(seed, index)
        m = ref_model.ModelTraj(points=torch.from_numpy(pts), wps_poses=torch.from_numpy(poses), wps_quats=torch.from_numpy(quats),
                                intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU, min_dist=clip[0], max_dist=clip[1])
        m(vis_wps_dist=0.0)
        m.loss["vis"].backward()
        save(f"traj_stress_{seed}_{index}", recipe=np.asarray("stress"), seed=seed, index=index, what=np.asarray(what), poses=poses, quats=quats,
             min_dist=np.float64(clip[0]), max_dist=np.float64(clip[1]), points_checksum=np.float64(pts.astype(np.float64).sum()),
             loss_vis=m.loss["vis"], rewards=m.rewards, vis_poses_grad=m.poses.grad, vis_quats_grad=m.quats.grad)
        g = m.poses.grad.numpy()
        print(f"   {what}: loss {float(m.loss['vis']):.6g}, NaN rewards {int(torch.isnan(m.rewards).sum())} of {len(pts)}, "
              f"NaN gradient rows {np.flatnonzero(np.isnan(g).any(axis=1)).tolist()}, largest row {np.nanmax(np.abs(g)) if np.isfinite(g).any() else float('nan'):.3g}")


def gen_full():
    """The reference itself at the bench cloud's full size: 1 M points (the BASELINE slab, seed 0) x 16 waypoints, fwd + bwd on the
    CPU (~2 GB of autograd state, a few seconds).  Stored: loss, the gradients, every 997th reward and the rewards' f64 sum — the
    cloud by recipe + checksum.  python tests/golden/make_golden.py full"""
    n, w = 1_000_000, 16
    pts = synth.make_cloud(n, seed=0)
    poses, quats = synth.make_path(w, optical=True)
    m = ref_model.ModelTraj(points=torch.from_numpy(pts), wps_poses=torch.from_numpy(poses), wps_quats=torch.from_numpy(quats),
                            intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU, min_dist=1.0, max_dist=5.0)
    m(vis_wps_dist=0.0)
    m.loss["vis"].backward()
    r = m.rewards.detach().numpy()
    save("traj_full_1m_16", recipe=np.asarray("room"), n=n, seed=0, extent=np.asarray((40.0, 40.0, 4.0)), poses=poses, quats=quats,
         min_dist=np.float64(1.0), max_dist=np.float64(5.0), points_checksum=np.float64(pts.astype(np.float64).sum()),
         loss_vis=m.loss["vis"], rewards_every_997th=r[::997].copy(), rewards_sum=np.float64(r.astype(np.float64).sum()),
         rewards_above_half=np.int64((r > 0.5).sum()), vis_poses_grad=m.poses.grad, vis_quats_grad=m.quats.grad)


def gen_timing():
    """Not a fixture: wall time of the reference itself (torch CPU, this container) on the bench workload's shape, for
    the record kept in profiles/r01_reference_cpu_timing.txt.  python tests/golden/make_golden.py timing"""
    import time
    print(f"reference ModelTraj / hidden_pts_removal on torch {torch.__version__} CPU, {torch.get_num_threads()} threads, "
          f"{os.cpu_count()} vCPUs")
    for n, w in ((100_000, 32), (1_000_000, 16)):
        pts = synth.make_cloud(n, seed=0)
        poses, quats = synth.make_path(w, optical=True)
        best = None
        for rep in range(3):
            m = ref_model.ModelTraj(points=torch.from_numpy(pts), wps_poses=torch.from_numpy(poses),
                                    wps_quats=torch.from_numpy(quats), intrins=K, img_width=IMG_W, img_height=IMG_H, device=CPU)
            t0 = time.perf_counter()
            m(vis_wps_dist=0.0)
            t1 = time.perf_counter()
            m.loss["vis"].backward()
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1)
        print(f"ModelTraj {n} points x {w} waypoints: fwd {best[1]:.3f} s, bwd {best[2]:.3f} s, fwd+bwd {best[0]:.3f} s "
              f"= {n * w / best[0]:.3e} evals/s (best of 3)")
    for n in (100_000, 1_000_000):
        pts = torch.from_numpy(synth.make_cloud(n, seed=0))
        t0 = time.perf_counter()
        vis, _ = ref_tools.hidden_pts_removal(pts, CPU)
        print(f"hidden_pts_removal {n} points: {time.perf_counter() - t0:.3f} s ({vis.shape[0]} visible)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["traj", "ties", "adam", "pose", "funcs", "hard", "ingest"]
    for w in which:
        globals()["gen_" + w]()
