"""ctypes binding of the CPU oracle (oracle/vis_oracle.c) + the numpy/scipy HPR oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by trajectory_optimization_amd/.  Pinned against the golden vectors
in tests/golden/ (generated from the reference by tests/golden/make_golden.py) by
tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")   # (ORACLE_LIB: the sanitizer build, oracle/Makefile)
_lib = None


def build(force=False):
    if os.environ.get("ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vis_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


_DT = {"f32": np.float32, "f64": np.float64}
c_f, c_d, c_i64, c_int = ctypes.c_float, ctypes.c_double, ctypes.c_int64, ctypes.c_int


def traj_forward(points, poses, quats, K, img_w, img_h, min_dist=1.0, max_dist=5.0, prec="f32", occ=None):
    """Visibility term of ModelTraj.forward over the given (already subsampled) waypoints.
    occ: optional (W,N) 0/1 occlusion masks multiplied into p per waypoint."""
    pts, poses, quats, K = _f32(points), _f32(poses), _f32(quats), _f32(K)
    occ = _f32(occ) if occ is not None else None
    N, W = pts.shape[0], poses.shape[0]
    dt = _DT[prec]
    lo, rew = np.empty(N, dt), np.empty(N, dt)
    pmin, pmax = np.empty(W, dt), np.empty(W, dt)
    mean, loss = c_d(), c_d()
    fn = getattr(lib(), "oracle_traj_forward_" + prec)
    rc = fn(_ptr(pts), c_i64(N), _ptr(poses), _ptr(quats), c_i64(W), _ptr(K), c_f(img_w), c_f(img_h),
            c_f(min_dist), c_f(max_dist), _ptr(occ), _ptr(lo), _ptr(rew), _ptr(pmin), _ptr(pmax),
            ctypes.byref(mean), ctypes.byref(loss))
    assert rc == 0
    return dict(lo_sum=lo, rewards=rew, pmin=pmin, pmax=pmax, mean_reward=mean.value, loss_vis=loss.value, occ=occ)


def traj_backward(points, poses, quats, K, img_w, img_h, fwd, gout=1.0, min_dist=1.0, max_dist=5.0, prec="f32", act_shift=0.0, phat_shift=0.0):
    """act_shift != 0 (diagnostic): the lower activity threshold of the clipped log-odds at 1/2 + act_shift instead of 1/2 —
    the difference of two such gradients is what the points inside the band are worth (tests/test_hip_reference_dense.py).
    phat_shift != 0 (diagnostic): every p_hat of the backward moved by that much (tools/stress_models.py)."""
    pts, poses, quats, K = _f32(points), _f32(poses), _f32(quats), _f32(K)
    N, W = pts.shape[0], poses.shape[0]
    dt = _DT[prec]
    pg, qg = np.empty((W, 3), dt), np.empty((W, 4), dt)
    rew = np.ascontiguousarray(fwd["rewards"], dtype=dt)
    fn = getattr(lib(), "oracle_traj_backward_" + prec)
    setter = getattr(lib(), "oracle_set_act_shift_" + prec)
    setter.argtypes = [c_d]
    setter(float(act_shift))
    setter2 = getattr(lib(), "oracle_set_phat_shift_" + prec)
    setter2.argtypes = [c_d]
    setter2(float(phat_shift))
    try:
        rc = fn(_ptr(pts), c_i64(N), _ptr(poses), _ptr(quats), c_i64(W), _ptr(K), c_f(img_w), c_f(img_h),
                c_f(min_dist), c_f(max_dist), _ptr(fwd.get("occ")), _ptr(rew), c_d(fwd["mean_reward"]), c_d(gout), _ptr(pg),
                _ptr(qg))
    finally:
        setter(0.0)
        setter2(0.0)
    assert rc == 0
    return pg, qg


# ---- the same path cut where a point-sharded run cuts it (tests/test_distributed_cpu.py: the CPU rehearsal of distributed.PointShard)

def traj_extrema(points, poses, quats, K, img_w, img_h, min_dist=1.0, max_dist=5.0, prec="f64"):
    """-> (min p, max p) per waypoint over THESE points (ranks combine them with min / max)."""
    pts, poses, quats, K = _f32(points), _f32(poses), _f32(quats), _f32(K)
    W = poses.shape[0]
    pmin, pmax = np.empty(W, _DT[prec]), np.empty(W, _DT[prec])
    fn = getattr(lib(), "oracle_traj_extrema_" + prec)
    assert fn(_ptr(pts), c_i64(pts.shape[0]), _ptr(poses), _ptr(quats), c_i64(W), _ptr(K), c_f(img_w), c_f(img_h), c_f(min_dist), c_f(max_dist),
              _ptr(pmin), _ptr(pmax)) == 0
    return pmin, pmax


def traj_forward_ext(points, poses, quats, K, img_w, img_h, ext_min, ext_max, min_dist=1.0, max_dist=5.0, prec="f64"):
    """-> (lo_sum, rewards) of these points for GIVEN per-waypoint extrema (min p, max p)."""
    pts, poses, quats, K = _f32(points), _f32(poses), _f32(quats), _f32(K)
    dt = _DT[prec]
    lo, rew = np.empty(pts.shape[0], dt), np.empty(pts.shape[0], dt)
    emin, emax = np.ascontiguousarray(ext_min, dtype=dt), np.ascontiguousarray(ext_max, dtype=dt)
    fn = getattr(lib(), "oracle_traj_forward_ext_" + prec)
    assert fn(_ptr(pts), c_i64(pts.shape[0]), _ptr(poses), _ptr(quats), c_i64(poses.shape[0]), _ptr(K), c_f(img_w), c_f(img_h), c_f(min_dist),
              c_f(max_dist), _ptr(emin), _ptr(emax), _ptr(lo), _ptr(rew)) == 0
    return lo, rew


def traj_backward_partial(points, poses, quats, K, img_w, img_h, ext_min, ext_max, rewards, min_dist=1.0, max_dist=5.0, prec="f64"):
    """-> (W, 40) float64: these points' sums per waypoint with dL/d reward = 1 (additive over parts of the cloud)."""
    pts, poses, quats, K = _f32(points), _f32(poses), _f32(quats), _f32(K)
    dt = _DT[prec]
    W = poses.shape[0]
    out = np.empty((W, 40), np.float64)
    emin, emax, rew = (np.ascontiguousarray(a, dtype=dt) for a in (ext_min, ext_max, rewards))
    fn = getattr(lib(), "oracle_traj_backward_partial_" + prec)
    assert fn(_ptr(pts), c_i64(pts.shape[0]), _ptr(poses), _ptr(quats), c_i64(W), _ptr(K), c_f(img_w), c_f(img_h), c_f(min_dist), c_f(max_dist),
              _ptr(emin), _ptr(emax), _ptr(rew), _ptr(out)) == 0
    return out


def traj_backward_final(poses, quats, partial, scale, prec="f64"):
    """The parts' sums added up -> (poses_grad, quats_grad); scale = dL/d reward (-gout vis^2 / N_all)."""
    poses, quats = _f32(poses), _f32(quats)
    W = poses.shape[0]
    dt = _DT[prec]
    pg, qg = np.empty((W, 3), dt), np.empty((W, 4), dt)
    part = np.ascontiguousarray(partial, dtype=np.float64)
    fn = getattr(lib(), "oracle_traj_backward_final_" + prec)
    assert fn(_ptr(poses), _ptr(quats), c_i64(W), _ptr(part), c_d(float(scale)), _ptr(pg), _ptr(qg)) == 0
    return pg, qg


def pose_forward(points, trans, quat, K, img_w, img_h, min_dist=1.0, max_dist=5.0, mask=None, prec="f32"):
    pts, trans, quat, K = _f32(points), _f32(trans).reshape(3), _f32(quat).reshape(4), _f32(K)
    mask = _f32(mask) if mask is not None else None
    N = pts.shape[0]
    obs = np.empty(N, _DT[prec])
    loss = c_d()
    fn = getattr(lib(), "oracle_pose_forward_" + prec)
    rc = fn(_ptr(pts), c_i64(N), _ptr(trans), _ptr(quat), _ptr(K), c_f(img_w), c_f(img_h), c_f(min_dist),
            c_f(max_dist), _ptr(mask), _ptr(obs), ctypes.byref(loss))
    assert rc == 0
    return obs, loss.value


def pose_backward(points, trans, quat, K, img_w, img_h, loss, gout=1.0, min_dist=1.0, max_dist=5.0, mask=None,
                  prec="f32"):
    pts, trans, quat, K = _f32(points), _f32(trans).reshape(3), _f32(quat).reshape(4), _f32(K)
    mask = _f32(mask) if mask is not None else None
    dt = _DT[prec]
    tg, qg = np.empty(3, dt), np.empty(4, dt)
    fn = getattr(lib(), "oracle_pose_backward_" + prec)
    rc = fn(_ptr(pts), c_i64(pts.shape[0]), _ptr(trans), _ptr(quat), _ptr(K), c_f(img_w), c_f(img_h),
            c_f(min_dist), c_f(max_dist), _ptr(mask), c_d(loss), c_d(gout), _ptr(tg), _ptr(qg))
    assert rc == 0
    return tg.reshape(1, 3), qg.reshape(1, 4)


def to_camera_frame(points, quat, trans, normalize=True, prec="f32"):
    pts, trans, quat = _f32(points), _f32(trans).reshape(3), _f32(quat).reshape(4)
    out = np.empty((pts.shape[0], 3), _DT[prec])
    fn = getattr(lib(), "oracle_to_camera_frame_" + prec)
    assert fn(_ptr(pts), c_i64(pts.shape[0]), _ptr(quat), _ptr(trans), c_int(int(normalize)), _ptr(out)) == 0
    return out


def soft_masks(cam_points, K, img_w, img_h, min_dist=1.0, max_dist=5.0, prec="f32"):
    pts, K = _f32(cam_points), _f32(K)
    N = pts.shape[0]
    d, f = np.empty(N, _DT[prec]), np.empty(N, _DT[prec])
    fn = getattr(lib(), "oracle_soft_masks_" + prec)
    assert fn(_ptr(pts), c_i64(N), _ptr(K), c_f(img_w), c_f(img_h), c_f(min_dist), c_f(max_dist), _ptr(d), _ptr(f)) == 0
    return d, f


def frustum_masks(cam_3xN, K, img_w, img_h, min_dist=1.0, max_dist=10.0):
    """get_cam_frustum_pts masks on a (3,N) camera-frame array -> (dist_mask, fov_mask) bool."""
    pts, K = _f32(cam_3xN), _f32(K)
    N = pts.shape[1]
    d, f = np.empty(N, np.uint8), np.empty(N, np.uint8)
    assert lib().oracle_frustum_masks(_ptr(pts), c_i64(N), _ptr(K), c_f(img_w), c_f(img_h), c_f(min_dist),
                                      c_f(max_dist), _ptr(d), _ptr(f)) == 0
    return d.astype(bool), f.astype(bool)


def spherical_flip(points, param=2):
    pts = _f32(points)
    out = np.empty_like(pts)
    rad = c_f()
    assert lib().oracle_spherical_flip(_ptr(pts), c_i64(pts.shape[0]), c_d(float(param)), _ptr(out),
                                       ctypes.byref(rad)) == 0
    return out, rad.value


def occlusion_masks(points, poses, quats, K, img_w, img_h, min_dist=1.0, max_dist=15.0, param=2):
    """Per-waypoint occlusion masks (W,N) f32 by the reference's hard per-camera pipeline
    (/root/reference/src/pc_processor.py:158-187): to_camera_frame (normalised quaternion, model.py:50-57) ->
    get_cam_frustum_pts -> hidden_pts_removal of the kept points seen from the camera centre.  A point inside the
    hard frustum that HPR hides gets 0; everything else 1 (the soft weights of the model already fade it out)."""
    pts = _f32(points)
    W, N = len(poses), pts.shape[0]
    occ = np.ones((W, N), np.float32)
    for w in range(W):
        cam = to_camera_frame(pts, quats[w], poses[w], normalize=True)
        d, f = frustum_masks(np.ascontiguousarray(cam.T), K, img_w, img_h, min_dist, max_dist)
        kept = np.flatnonzero(d & f)
        if len(kept) >= 4:
            vis, _ = hidden_pts_removal(cam[kept], param)
            hidden = np.ones(len(kept), bool)
            hidden[vis] = False
            occ[w, kept[hidden]] = 0.0
    return occ


def hidden_pts_removal(points, param=2):
    """/root/reference/src/tools.py:56-85: flip, append the origin, Qhull (scipy.spatial.ConvexHull —
    the same third-party hull the reference calls, scipy pin 1.5.4 in /root/reference/requirements.txt:11),
    drop the LAST hull vertex unconditionally (quirk Q3), scatter a 0/1 mask.
    Returns (visible_idx int64 ascending, mask f32)."""
    from scipy.spatial import ConvexHull
    flipped, _ = spherical_flip(points, param)
    hull = ConvexHull(np.concatenate([flipped, np.zeros((1, 3), np.float32)], axis=0))
    visible = hull.vertices[:-1]
    mask = np.zeros(len(flipped), np.float32)
    mask[visible] = 1
    return np.asarray(visible, dtype=np.int64), mask
