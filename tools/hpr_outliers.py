"""Repeated HPR builds of one cloud, every call timed: outliers (a retry with a larger face pool, a careful round) show up here."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_optimization_amd import synth, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
P = torch.from_numpy(synth.make_cloud(n, seed=0)).to("cuda:0")
ops.hidden_pts_removal(P)
ts = []
for k in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    idx, _ = ops.hidden_pts_removal(P)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"call {k}: {ts[-1]:.2f} ms visible {idx.numel()}", file=sys.stderr)
ts.sort()
print(f"n={n}: min {ts[0]:.2f} median {ts[len(ts)//2]:.2f} p90 {ts[int(len(ts)*0.9)]:.2f} max {ts[-1]:.2f} ms")
