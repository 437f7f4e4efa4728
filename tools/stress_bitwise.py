"""Stress (run on the GPU box: python tools/stress_bitwise.py [n_configs]): random clouds / paths / rigs / clip limits; dense == culled == masked variants, bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_optimization_amd import synth, ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(123)
fails = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    n = int(rng.choice([700, 5000, 40_000, 140_000, 300_000, 600_000]))
    w = int(rng.integers(1, 140))
    cams = int(rng.choice([1, 1, 1, 2, 5]))
    if cams > 1: w = max(1, w // cams)
    scale = float(rng.choice([0.3, 1.0, 2.5]))
    pts = (synth.make_cloud(n, seed=int(rng.integers(1 << 30))) * np.float32(scale)).astype(np.float32)
    if rng.random() < 0.3: pts = np.concatenate([pts, pts[: n // 5]])  # ties
    poses, quats = synth.make_path(w, optical=True, jitter_seed=int(rng.integers(1 << 30)))
    quats = (quats * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
    clip = (float(rng.uniform(0.2, 2.0)), float(rng.uniform(3.0, 12.0)))
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P, sort=bool(rng.random() < 0.85))
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT, clip[0], clip[1])
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    rg = ops.CameraRig(*synth.camera_rig(cams), dev) if cams > 1 else None
    ws = ops.TrajWorkspace(cloud, w * cams)
    occ = ops.occlusion_bits(cloud, P, p, q, cam, 1.0, 15.0, "zbuffer") if (cams == 1 and rng.random() < 0.25) else None
    gout = torch.ones(1, device=dev)
    g = torch.rand(pts.shape[0], generator=torch.Generator().manual_seed(it)).to(dev) - 0.4
    outs = []
    for flags in (0, ops.DENSE):
        lo, mm, need = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags, occ=occ, want_need=True)
        rew, sc = ops.traj_reward(cloud, lo, cam, ws)
        for kw in (dict(scalars=sc, gout=gout), dict(grad_rewards=g)):
            a = ops.traj_backward(cloud, p, q, cam, ws, lo, mm, rig=rg, flags=flags, occ=occ, **kw)
            b = ops.traj_backward(cloud, p, q, cam, ws, lo, mm, rig=rg, flags=flags, occ=occ, need_mask=need, **kw)
            outs.append((lo, mm, rew, a[0], a[1], b[0], b[1]))
        if flags:
            scan = ops.traj_backward_scan(cloud, p, q, cam, ws, mm, rig=rg, flags=flags, occ=occ)
            c = ops.traj_backward(cloud, p, q, cam, ws, lo, mm, rig=rg, flags=flags, occ=occ, need_mask=scan, scalars=sc, gout=gout)
            outs.append((lo, mm, rew, c[0], c[1], c[0], c[1]))
    ref = outs[0]
    def same(x, y): return torch.equal(x, y) or (torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))
    ok = all(same(o[0], ref[0]) and same(o[1], ref[1]) and same(o[2], ref[2]) for o in outs)
    ok &= all(same(o[3], o[5]) and same(o[4], o[6]) for o in outs)                 # masked == unmasked
    ok &= same(outs[0][3], outs[2][3]) and same(outs[0][4], outs[2][4])           # culled == dense (fused loss)
    ok &= same(outs[1][3], outs[3][3]) and same(outs[1][4], outs[3][4])           # culled == dense (grad_rewards)
    ok &= same(outs[4][3], outs[2][3])                                             # scan path == fused
    if not ok:
        names = ["culled/loss", "culled/grad", "dense/loss", "dense/grad", "dense/scan"]
        for nm, o in zip(names, outs):
            print("   ", nm, "fwd==ref", same(o[0], ref[0]), same(o[1], ref[1]), same(o[2], ref[2]), "masked==plain", same(o[3], o[5]), same(o[4], o[6]),
                  "max|d| plain-vs-ref", float((o[3] - outs[0][3]).abs().max()), "masked-vs-plain", float((o[3] - o[5]).abs().max()),
                  "a>0 wps", int((o[1][:, 0] > 0).sum()))
        fails += 1
        print("MISMATCH", it, n, w, cams, scale, clip, occ is not None)
print("stress done, failures:", fails)
