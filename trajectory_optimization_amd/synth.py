"""Seeded synthetic inputs for the BASELINE.json configurations.

One definition shared by the golden-fixture generator, the parity tests and
bench.py so that every leg sees bit-identical inputs.  numpy's PCG64 stream is
stable across numpy versions, so the inputs do not depend on the torch build.

Shapes follow the reference's input contract: cloud (N,3) f32 as produced by
its PointCloud2 ingest (/root/reference/src/trajectory_optimization.py:62-63),
waypoint positions (W,3) f32 and wxyz quaternions (W,4) f32
(/root/reference/src/trajectory_optimization.py:66-80), intrinsics as returned
by load_intrinsics (/root/reference/src/tools.py:320-325).
"""
import numpy as np

# /root/reference/src/tools.py:320-325
IMG_WIDTH = 1232.0
IMG_HEIGHT = 1616.0
K_INTRINS = np.array([[758.03967, 0.0, 621.46572],
                      [0.0, 761.62359, 756.86402],
                      [0.0, 0.0, 1.0]], dtype=np.float32)

# body -> optical frame (camera +Z along body +X, +X to the right, +Y down), wxyz
Q_OPTICAL = np.array([0.5, -0.5, 0.5, -0.5], dtype=np.float64)


def quat_mul(a, b):
    """Hamilton product, wxyz, broadcasting over leading dims (float64)."""
    aw, ax, ay, az = np.moveaxis(np.asarray(a, dtype=np.float64), -1, 0)
    bw, bx, by, bz = np.moveaxis(np.asarray(b, dtype=np.float64), -1, 0)
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def make_cloud(n_points, seed=0, extent=(40.0, 40.0, 4.0)):
    """Uniform slab `rand(N,3)*extent - extent/2` metres, f32."""
    rng = np.random.default_rng(seed)
    ext = np.asarray(extent, dtype=np.float64)
    pts = rng.random((n_points, 3)) * ext - ext / 2.0
    return pts.astype(np.float32)


def make_path(n_wps, optical=True, jitter_seed=None, scale=1.0):
    """Smooth curve t=scale*(-10+20s, 3 sin 6s, 0), yaw 0.5 cos 6s, wxyz quats.

    With `optical` the yaw rotation is composed with the body->optical
    rotation so the camera +Z axis looks along the path.  `jitter_seed`
    adds a small random roll/pitch and a non-unit scale to the quaternions
    (exercises the F.normalize step of to_camera_frame).
    """
    if n_wps == 1:
        s = np.zeros(1)
    else:
        s = np.arange(n_wps, dtype=np.float64) / (n_wps - 1)
    pos = scale * np.stack([-10.0 + 20.0 * s, 3.0 * np.sin(6.0 * s), np.zeros_like(s)], axis=1)
    yaw = 0.5 * np.cos(6.0 * s)
    q = np.stack([np.cos(yaw / 2), np.zeros_like(s), np.zeros_like(s), np.sin(yaw / 2)], axis=1)
    if optical:
        q = quat_mul(q, Q_OPTICAL[None, :])
    if jitter_seed is not None:
        rng = np.random.default_rng(jitter_seed)
        dq = np.concatenate([np.ones((n_wps, 1)), 0.05 * rng.standard_normal((n_wps, 3))], axis=1)
        q = quat_mul(q, dq) * (0.5 + rng.random((n_wps, 1)))
    return pos.astype(np.float32), q.astype(np.float32)


def camera_rig(n_cams=5):
    """Fixed extrinsics of a multi-camera rig: yaw offsets 0, +-72, +-144 deg
    about the body z axis, shared K (BASELINE.json config 5). Returns wxyz
    quaternions (C,4) f32 rotating rig->body and zero lever arms (C,3)."""
    offs = np.deg2rad(np.array([0.0, 72.0, -72.0, 144.0, -144.0][:n_cams]))
    q = np.stack([np.cos(offs / 2), np.zeros_like(offs), np.zeros_like(offs), np.sin(offs / 2)], axis=1)
    return q.astype(np.float32), np.zeros((n_cams, 3), dtype=np.float32)


def with_duplicate_rows(base, dup, seed):
    """`base` (n,3) plus copies of its rows `dup` (and of dup[:len(dup)//6] once more), all rows then shuffled (PCG64(seed)): a cloud
    with exact duplicate rows whose copies sit before and after their originals.  -> (points, source row of each point)."""
    src = np.concatenate([np.arange(len(base)), np.asarray(dup), np.asarray(dup)[:len(dup) // 6]])
    order = np.random.default_rng(seed).permutation(len(src))
    src = src[order]
    return base[src].astype(np.float32), src
