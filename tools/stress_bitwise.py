"""Stress (run on the GPU box: python tools/stress_bitwise.py [n_configs]): random clouds / paths / rigs / clip limits / tie
sets; the dense and the culled mode, both loss heads, and the multi-trajectory entry points against one-by-one runs,
bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_optimization_amd import synth, ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(123)


def same(x, y):
    return torch.equal(x, y) or bool((torch.isnan(x) == torch.isnan(y)).all()) and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))


def one(cloud, p, q, cam, rg, flags, occ, g):
    """forward -> reward -> backward (fused loss head and grad_rewards head) of one trajectory"""
    w = p.shape[0]
    ws = ops.TrajWorkspace(cloud, w * (rg.n_cams if rg else 1))
    half = torch.empty(cloud.n, device=dev)
    lo, mm = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags, occ=occ, rewards_half=half)
    rew, sc = ops.traj_reward(cloud, lo, cam, ws, rewards=half, prefilled=True)
    a = ops.traj_backward(cloud, w, cam, ws, lo, scalars=sc, gout=torch.ones(1, device=dev), rig=rg, flags=flags, occ=occ)
    lo2, _ = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags, occ=occ)
    b = ops.traj_backward(cloud, w, cam, ws, lo2, grad_rewards=g, rig=rg, flags=flags, occ=occ)
    return [lo[:cloud.n].clone(), mm, rew, sc[:2].clone(), a[0], a[1], b[0], b[1]]


fails = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    n = int(rng.choice([700, 5000, 40_000, 140_000, 300_000, 600_000]))
    w = int(rng.integers(1, 140))
    cams = int(rng.choice([1, 1, 1, 2, 5]))
    if cams > 1: w = max(1, w // cams)
    scale = float(rng.choice([0.3, 1.0, 2.5]))
    pts = (synth.make_cloud(n, seed=int(rng.integers(1 << 30))) * np.float32(scale)).astype(np.float32)
    if rng.random() < 0.3: pts = np.concatenate([pts, pts[: n // 5], pts[: n // 9]])  # ties (2- and 3-fold)
    poses, quats = synth.make_path(w, optical=True, jitter_seed=int(rng.integers(1 << 30)))
    quats = (quats * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
    clip = (float(rng.uniform(0.2, 2.0)), float(rng.uniform(3.0, 12.0)))
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P, sort=bool(rng.random() < 0.85))
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT, clip[0], clip[1])
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    rg = ops.CameraRig(*synth.camera_rig(cams), dev) if cams > 1 else None
    occ = ops.occlusion_bits(cloud, P, p, q, cam, 1.0, 15.0, "zbuffer") if (cams == 1 and rng.random() < 0.25) else None
    g = torch.rand(pts.shape[0], generator=torch.Generator().manual_seed(it)).to(dev) - 0.4
    culled, dense = one(cloud, p, q, cam, rg, 0, occ, g), one(cloud, p, q, cam, rg, ops.DENSE, occ, g)
    again = one(cloud, p, q, cam, rg, 0, occ, g)
    ok = all(same(x, y) for x, y in zip(culled, dense)) and all(same(x, y) for x, y in zip(culled, again))
    what = "dense/culled/repeat"
    if ok and occ is None and w >= 2:
        # the same waypoints as two trajectories in one pass
        cut = int(rng.integers(1, w))
        toff = torch.tensor([0, cut, w], dtype=torch.int32, device=dev)
        V = w * cams
        wsm = ops.TrajWorkspace(cloud, V, 2)
        lom, mmm = ops.traj_forward_multi(cloud, p, q, toff, cam, wsm, rg)
        rewm, scm = ops.traj_reward_multi(cloud, lom, cam, wsm)
        pgm, qgm = ops.traj_backward_multi(cloud, w, 2, cam, wsm, lom, scalars=scm, gout=torch.ones(2, device=dev), rig=rg)
        for k, (lo_, hi_) in enumerate(((0, cut), (cut, w))):
            s = one(cloud, p[lo_:hi_].contiguous(), q[lo_:hi_].contiguous(), cam, rg, 0, None, g)
            ok &= same(s[0], lom[k, :cloud.n]) and same(s[2], rewm[k]) and same(s[3], scm[k, :2]) and same(s[4], pgm[lo_:hi_]) and same(s[5], qgm[lo_:hi_])
        what = "multi vs one by one"
    if not ok:
        fails += 1
        print("MISMATCH", what, it, n, w, cams, scale, clip, occ is not None)
print("stress done, failures:", fails)
