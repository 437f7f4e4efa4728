"""HPR timing on the GPU next to scipy/Qhull on the host (same flipped input)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_optimization_amd import synth, ops
from oracle import oracle
dev = torch.device("cuda:0")
for n in (10_000, 100_000, 1_000_000):
    pts = synth.make_cloud(n, seed=0)
    P = torch.from_numpy(pts).to(dev)
    ops.hidden_pts_removal(P)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        idx, _ = ops.hidden_pts_removal(P)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    _, rounds = ops.hull_vertices_with_origin(ops.spherical_flip(P)[0], True, return_rounds=True)
    t = time.perf_counter(); ref, _ = oracle.hidden_pts_removal(pts); dq = time.perf_counter() - t
    print(f"n={n} visible={idx.numel()} gpu_ms={dt*1e3:.2f} rounds={rounds} qhull_ms={dq*1e3:.1f} equal={np.array_equal(idx.cpu().numpy().astype(np.int64), ref)}")
