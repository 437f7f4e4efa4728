"""The batched HPR of the occlusion workload alone (128 camera views of a 1 M-point cloud), a few repeats: for
rocprofv3 --kernel-trace --stats."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import ops, synth
W, N = 128, 1_000_000
dev = torch.device("cuda:0")
pts = torch.from_numpy(synth.make_cloud(N, seed=0)).to(dev)
poses, quats = synth.make_path(W, optical=True)
poses, quats = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
segs = []
for w in range(W):
    c3 = ops.to_camera_frame_exact(pts, quats[w], poses[w], normalize=True, transpose=True)
    _, _, idx = ops.frustum_cull(c3, cam, 1.0, 15.0)
    segs.append(c3[:, idx.long()].t().contiguous())
offs = np.concatenate([[0], np.cumsum([s.shape[0] for s in segs])])
allp = torch.cat(segs)
ops.hidden_pts_removal_batched(allp, offs)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ops.hidden_pts_removal_batched(allp, offs)
torch.cuda.synchronize(); print(f"batched HPR, {W} segments, {allp.shape[0]} points: {1e3 * (time.perf_counter() - t0) / reps:.1f} ms")
