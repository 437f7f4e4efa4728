"""The 1e-5 parity bar is conditional, and the condition is stated here: the clipped log-odds (/root/reference/src/model.py:229)
has a gradient jump at p_hat = 1/2 and at p_hat = 1 - 1e-6, so a point within f32 rounding of a threshold contributes to its
waypoint's gradient or not depending on the last bit of p_hat — the reference's own f32 result is as undecided.  Over 40 seeded
random configurations every waypoint whose nearest point keeps a margin from both thresholds must meet the bar; the number of
waypoints excluded is reported (a warning in the test summary), not hidden."""
import warnings

import numpy as np
import pytest
import torch

from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
MARGIN = 3e-7     # |p_hat - threshold| below this is inside f32 rounding of p_hat (2^-23 relative, a few operations)
GRAD_TOL = 1e-5   # north star


def _margins(pts, poses, quats, clip):
    """f64 restatement of p_hat per waypoint (model.py:13-57, 223-227): distance of the nearest point to each threshold."""
    P = pts.astype(np.float64)
    mean, std = np.float64(np.float32((clip[0] + clip[1]) / 2)), np.float64(np.float32((clip[1] - clip[0]) / 2))
    hi = np.float64(np.float32(1 - 1e-6))
    out = []
    for pose, q in zip(poses.astype(np.float64), quats.astype(np.float64)):
        w, x, y, z = q / np.linalg.norm(q)
        R = np.array([[w*w+x*x-y*y-z*z, 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), w*w-x*x+y*y-z*z, 2*(y*z-w*x)],
                      [2*(x*z-w*y), 2*(y*z+w*x), w*w-x*x-y*y+z*z]])
        C = (P - pose) @ R
        H = C @ K.astype(np.float64).T
        zz = H[:, 2] + 1e-6
        au, av = (H[:, 0] / zz - IW / 2) / IW, (H[:, 1] / zz - IH / 2) / IH
        p = np.exp(-0.5 * (np.linalg.norm(C - mean, axis=1) / std) ** 2 - 0.5 * au * au - 0.5 * av * av) / (1 + np.exp(-H[:, 2]))
        ph = (p - p.min()) / (p - p.min()).max()
        below = ph[ph < 1]
        out.append(min(np.abs(ph - 0.5).min(), np.abs(below - hi).min() if below.size else 1.0))
    return np.array(out)


def configurations(count=40, seed=2026):
    """The seeded random configurations of the test below (tests/golden/make_golden.py `dense` runs the reference on one of them)
    -> (index, points, poses, quats, clip limits, dense flag)."""
    rng = np.random.default_rng(seed)
    for it in range(count):
        n = int(rng.choice([900, 6000, 30_000, 90_000]))
        w = int(rng.integers(3, 24))
        scale = float(rng.choice([0.3, 1.0, 2.0]))
        pts = (synth.make_cloud(n, seed=int(rng.integers(1 << 30))) * np.float32(scale)).astype(np.float32)
        poses, quats = synth.make_path(w, optical=True, jitter_seed=int(rng.integers(1 << 30)))
        quats = (quats * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
        clip = (float(rng.uniform(0.3, 2.0)), float(rng.uniform(3.0, 10.0)))
        yield it, pts, poses, quats, clip, bool(rng.random() < 0.5)


def stress_configurations(count, seed):
    """The configurations of tools/stress_models.py [count] [seed], in its order -> (index, points, poses, quats, clip limits, dense
    flag, waypoint of the ModelPose check).  tests/golden/make_golden.py `stress` runs the reference on the ones listed in
    STRESS_CASES: what that stress run found outside the bars, each a property of the reference's own f32 arithmetic."""
    rng = np.random.default_rng(seed)
    for it in range(count):
        n = int(rng.choice([900, 6000, 30_000, 90_000]))
        w = int(rng.integers(3, 24))
        scale = float(rng.choice([0.3, 1.0, 2.0]))
        pts = (synth.make_cloud(n, seed=int(rng.integers(1 << 30))) * np.float32(scale)).astype(np.float32)
        poses, quats = synth.make_path(w, optical=True, jitter_seed=int(rng.integers(1 << 30)))
        quats = (quats * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
        clip = (float(rng.uniform(0.3, 2.0)), float(rng.uniform(3.0, 10.0)))
        dense = bool(rng.random() < 0.5)
        yield it, pts, poses, quats, clip, dense, int(rng.integers(0, w))


# (seed, index) of tools/stress_models.py -> what it is a case of
STRESS_CASES = {
    (23, 4): "amplification: 900 points, one active point 2.6e-4 below p_hat = 1 - 1e-6 carries a waypoint's gradient",
    (31, 83): "amplification: 6 000 points in 12 x 12 x 1.2 m",
    (31, 101): "cancellation: the whole gradient is 7e-9 (one point with reward 0.999999: r (1 - r) in f32)",
    (23, 134): "degenerate: every p of one waypoint underflows to 0 in f32 (0 / 0: NaN rewards, loss and gradients)",
}


def stress_case(seed, index):
    return next(c for c in stress_configurations(index + 1, seed) if c[0] == index)


def test_parity_bar_on_random_configurations_states_its_condition():
    from oracle import oracle
    from trajectory_optimization_amd.model import ModelTraj
    dev = torch.device("cuda:0")
    n_wps = n_excluded = 0
    worst = 0.0
    for it, pts, poses, quats, clip, dense in configurations():
        n, w = len(pts), len(poses)
        m = ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH,
                      min_dist=clip[0], max_dist=clip[1], device=dev, dense=dense)
        m(vis_wps_dist=0.0)
        m.loss["vis"].backward()
        f = oracle.traj_forward(pts, poses, quats, K, IW, IH, clip[0], clip[1], prec="f64")
        pg, qg = oracle.traj_backward(pts, poses, quats, K, IW, IH, f, min_dist=clip[0], max_dist=clip[1], prec="f64")
        # rewards and the loss are continuous in p_hat at both thresholds: no condition on them
        assert abs(m.loss["vis"].item() - f["loss_vis"]) <= 5e-6 * f["loss_vis"], it
        np.testing.assert_allclose(m.rewards.detach().cpu().numpy(), f["rewards"], rtol=1e-5, atol=0, err_msg=f"configuration {it}")
        keep = _margins(pts, poses, quats, clip) > MARGIN
        n_wps += w
        n_excluded += int((~keep).sum())
        gp, gq = m.poses.grad.cpu().numpy(), m.quats.grad.cpu().numpy()
        dp, dq = np.abs(pg).max(), np.abs(qg).max()   # (a configuration in which no point reaches p_hat >= 1/2 twice has zero gradient)
        ep = np.abs(gp - pg).max(axis=1) / (dp if dp > 0 else 1.0)
        eq = np.abs(gq - qg).max(axis=1) / (dq if dq > 0 else 1.0)
        assert (ep[keep] < GRAD_TOL).all() and (eq[keep] < GRAD_TOL).all(), (it, n, w, clip, ep.max(), eq.max())
        if keep.any():
            worst = max(worst, float(ep[keep].max()), float(eq[keep].max()))
    assert n_excluded <= n_wps // 20, (n_excluded, n_wps)   # the condition must stay the exception
    warnings.warn(f"threshold conditioning: {n_excluded} of {n_wps} waypoints (40 random configurations) have a point within {MARGIN:g} of "
                  f"p_hat = 1/2 or 1 - 1e-6 and were excluded from the 1e-5 gradient bar; worst error among the others {worst:.1e}")
