"""Where an occlusion refresh's time goes (1 M points x 128 waypoints, method hpr): cull | lay the kept clouds end to end | hull pass | bit rows."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import ops, synth
W, N = 128, 1_000_000
dev = torch.device("cuda:0")
P = torch.from_numpy(synth.make_cloud(N, seed=0)).to(dev)
poses, quats = synth.make_path(W, optical=True)
poses, quats = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
cloud = ops.PackedCloud(P) if hasattr(ops, "PackedCloud") else None
def sync(): torch.cuda.synchronize()
def timed(f, reps=5):
    f(); sync(); t = time.perf_counter()
    for _ in range(reps): f()
    sync(); return (time.perf_counter() - t) / reps * 1e3
res = {}
res["cull_waypoints"] = timed(lambda: ops.cull_waypoints(P, poses, quats, cam, 1.0, 15.0, normalize=True, scratch=True))
kept_all, pts_all, counts, kcnt = ops.cull_waypoints(P, poses, quats, cam, 1.0, 15.0, normalize=True, scratch=True)
offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
tot = int(offs[-1])
cat = torch.empty((tot, 3), dtype=torch.float32, device=dev)
def copies():
    for w in range(W):
        if counts[w]: cat[int(offs[w]):int(offs[w + 1])].copy_(pts_all[w, :counts[w]])
res["128 copies"] = timed(copies)
vis = torch.empty(tot, dtype=torch.float32, device=dev)
res["hull pass"] = timed(lambda: ops._hpr_batched_mask(cat, [int(o) for o in offs], vis))
if cloud is not None:
    res["occlusion_bits (all)"] = timed(lambda: ops.occlusion_bits(cloud, P, poses, quats, cam, 1.0, 15.0, "hpr"), reps=3)
for k, v in res.items(): print(f"{k:28s} {v:8.3f} ms")
