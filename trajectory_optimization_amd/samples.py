"""The reference's `.npz` sample format (/root/reference/src/trajectory_optimization_sample.py:29-50,
/root/reference/src/pose_optimization_sample.py): `point_cloud_<i>.npz` with key `pts` ((N,3) or (3,N)) and
`path_poses_<i>.npz` with key `poses` ((W,3)); orientations start as identity quaternions (wxyz)."""
import os

import numpy as np


def load_data(points_file, poses_file=None):
    """-> (pts (N,3) float32, poses (W,3) float32 or None, quats_wxyz (W,4) float32 or None)."""
    pts = np.load(points_file)["pts"]
    if pts.ndim != 2 or 3 not in pts.shape:
        raise ValueError(f"{points_file}: expected a (N,3) or (3,N) array under 'pts', got {pts.shape}")
    if pts.shape[1] > pts.shape[0]:
        pts = pts.T  # the samples store either layout; the long axis is N (trajectory_optimization_sample.py:36-40)
    if pts.shape[1] != 3:
        raise ValueError(f"{points_file}: expected a (N,3) or (3,N) array under 'pts', got {pts.shape}")
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    if poses_file is None:
        return pts, None, None
    poses = np.ascontiguousarray(np.load(poses_file)["poses"], dtype=np.float32)
    if poses.ndim != 2 or poses.shape[1] != 3:
        raise ValueError(f"{poses_file}: expected (W,3) under 'poses', got {poses.shape}")
    quats = np.tile(np.array([[1.0, 0.0, 0.0, 0.0]], dtype=np.float32), (len(poses), 1))
    return pts, poses, quats


def load_sequence(data_dir, index):
    """data/points/point_cloud_<index>.npz + data/paths/path_poses_<index>.npz under data_dir, as the reference lays them out."""
    return load_data(os.path.join(data_dir, "points", f"point_cloud_{index}.npz"),
                     os.path.join(data_dir, "paths", f"path_poses_{index}.npz"))


def save_result(path, poses, quats_wxyz, rewards=None, log=None):
    """Optimised trajectory (+ per-point rewards, + the gain log) as one .npz — what the reference publishes on ROS topics."""
    out = {"poses": np.asarray(poses, dtype=np.float32), "quats_wxyz": np.asarray(quats_wxyz, dtype=np.float32)}
    if rewards is not None:
        out["rewards"] = np.asarray(rewards, dtype=np.float32)
    for k, v in (log or {}).items():
        out["log_" + k] = np.asarray(v, dtype=np.float32)
    np.savez_compressed(path, **out)
