"""Stress (GPU box): hidden-point removal of random clouds of many shapes, batched and one by one, against scipy/Qhull.
python tools/stress_hpr.py [n_segments] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_optimization_amd import ops
from oracle import oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
dev = torch.device("cuda:0")
segs, kinds = [], []
for s in range(B):
    n = int(rng.choice([4, 5, 9, 40, 300, 2000, 9000]))
    kind = rng.choice(["ball", "shell", "slab", "cluster", "ring", "far", "tiny"])
    if kind == "ball":
        x = rng.normal(size=(n, 3)) * rng.uniform(0.5, 20)
    elif kind == "shell":
        u = rng.normal(size=(n, 3)); x = u / np.linalg.norm(u, axis=1, keepdims=True) * rng.uniform(2, 30) * (1 + 0.01 * rng.normal(size=(n, 1)))
    elif kind == "slab":
        x = rng.uniform(-1, 1, (n, 3)) * np.array([20, 20, 0.5]) + np.array([0, 0, rng.uniform(-5, 5)])
    elif kind == "cluster":
        c = rng.uniform(-10, 10, (4, 3)); x = c[rng.integers(0, 4, n)] + rng.normal(size=(n, 3)) * 0.3
    elif kind == "ring":
        t = rng.uniform(0, 2 * np.pi, n); x = np.stack([np.cos(t) * 8, np.sin(t) * 8, rng.normal(size=n) * 0.2], 1) + rng.normal(size=(n, 3)) * 0.05
    elif kind == "far":
        x = rng.normal(size=(n, 3)) * 2 + np.array([300.0, -150.0, 40.0])
    else:
        x = rng.normal(size=(n, 3)) * 1e-3 + 0.05
    segs.append(x.astype(np.float32)); kinds.append(kind)
offs = np.concatenate([[0], np.cumsum([len(s) for s in segs])])
idx, voff, mask, status = ops.hidden_pts_removal_batched(torch.from_numpy(np.concatenate(segs)).to(dev), offs)
idx, status = idx.cpu().numpy().astype(np.int64), status.cpu().numpy()
bad = 0
for s, pts in enumerate(segs):
    got = idx[voff[s]:voff[s + 1]] - offs[s]
    try:
        ref = oracle.hidden_pts_removal(pts)[0]
    except Exception as e:  # Qhull refuses (flat / too few): the GPU path must report a status
        if status[s] == 0:
            bad += 1; print("Qhull raised but status 0:", s, kinds[s], len(pts), type(e).__name__)
        continue
    if status[s] != 0 or not np.array_equal(got, ref):
        bad += 1
        print("MISMATCH", s, kinds[s], len(pts), "status", status[s], "gpu", len(got), "qhull", len(ref),
              "sym diff", len(np.setxor1d(got, ref)))
        continue
    if len(pts) >= 4:
        one = ops.hidden_pts_removal(torch.from_numpy(pts).to(dev))[0].cpu().numpy()
        if not np.array_equal(one, got):
            bad += 1; print("single != batched", s, kinds[s], len(pts))
print("hpr stress done:", B, "segments, failures:", bad)
