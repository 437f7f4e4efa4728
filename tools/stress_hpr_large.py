"""Stress (GPU box): hidden-point removal of LARGE random clouds of many shapes (the build's sample phase and its height-ranked
candidates on other geometry than the bench cloud), one by one and as one batch, against scipy/Qhull.
python tools/stress_hpr_large.py [n_clouds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_optimization_amd import ops
from oracle import oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
dev = torch.device("cuda:0")
segs, kinds = [], []
for s in range(B):
    n = int(rng.choice([40_000, 120_000, 400_000]))
    kind = rng.choice(["ball", "shell", "slab", "cluster", "ring", "far", "terrain"])
    if kind == "ball":
        x = rng.normal(size=(n, 3)) * rng.uniform(0.5, 20)
    elif kind == "shell":
        n = min(n, 120_000)   # every point visible: Qhull needs a while
        u = rng.normal(size=(n, 3)); x = u / np.linalg.norm(u, axis=1, keepdims=True) * rng.uniform(2, 30) * (1 + 0.01 * rng.normal(size=(n, 1)))
    elif kind == "slab":
        x = rng.uniform(-1, 1, (n, 3)) * np.array([20, 20, 0.5]) + np.array([0, 0, rng.uniform(-5, 5)])
    elif kind == "cluster":
        c = rng.uniform(-10, 10, (6, 3)); x = c[rng.integers(0, 6, n)] + rng.normal(size=(n, 3)) * 0.5
    elif kind == "ring":
        t = rng.uniform(0, 2 * np.pi, n); x = np.stack([np.cos(t) * 8, np.sin(t) * 8, rng.normal(size=n) * 0.2], 1) + rng.normal(size=(n, 3)) * 0.05
    elif kind == "far":
        x = rng.normal(size=(n, 3)) * 2 + np.array([300.0, -150.0, 40.0])
    else:
        xy = rng.uniform(-20, 20, (n, 2)); x = np.concatenate([xy, (np.sin(xy[:, :1] * 0.4) * np.cos(xy[:, 1:] * 0.3) * 1.5 - 2.0) + 0.02 * rng.normal(size=(n, 1))], 1)
    segs.append(x.astype(np.float32)); kinds.append(kind)
bad = 0
refs = []
for s, pts in enumerate(segs):
    ref = oracle.hidden_pts_removal(pts)[0]
    refs.append(ref)
    P = torch.from_numpy(pts).to(dev)
    ops.hidden_pts_removal(P)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = ops.hidden_pts_removal(P)[0]
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
    g = got.cpu().numpy().astype(np.int64)
    ok = np.array_equal(g, ref)
    bad += not ok
    print(f"{kinds[s]:8s} n={len(pts):7d} visible={len(ref):7d} {dt:7.2f} ms {'ok' if ok else 'MISMATCH: got %d, missing %d, extra %d' % (len(g), len(np.setdiff1d(ref, g)), len(np.setdiff1d(g, ref)))}")
offs = np.concatenate([[0], np.cumsum([len(s) for s in segs])])
allp = torch.from_numpy(np.concatenate(segs)).to(dev)
ops.hidden_pts_removal_batched(allp, offs)
torch.cuda.synchronize(); t0 = time.perf_counter()
idx, voff, mask, status = ops.hidden_pts_removal_batched(allp, offs)
torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
idx = idx.cpu().numpy().astype(np.int64)
for s in range(B):
    gb = idx[voff[s]:voff[s + 1]] - offs[s]
    if not np.array_equal(gb, refs[s]):
        bad += 1; print("batched MISMATCH", s, kinds[s], len(segs[s]), "got", len(gb), "ref", len(refs[s]), "missing", len(np.setdiff1d(refs[s], gb)), "extra", len(np.setdiff1d(gb, refs[s])), "status", int(status[s]))
print(f"batched: {B} clouds, {offs[-1]} points: {dt:.2f} ms; failures: {bad}")
