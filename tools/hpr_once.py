"""One warm-up and a few timed single-cloud HPR calls (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_optimization_amd import synth, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P = torch.from_numpy(synth.make_cloud(n, seed=0)).to("cuda:0")
ops.hidden_pts_removal(P)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps):
    idx, _ = ops.hidden_pts_removal(P)
torch.cuda.synchronize()
print(f"n={n} visible={idx.numel()} ms={(time.perf_counter() - t) / reps * 1e3:.2f}")
