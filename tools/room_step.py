"""The 10 x 10 x 4 m room of bench.py's density sweep, culled steps only: for rocprofv3 --kernel-trace --stats."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import ops, synth
dev = torch.device("cuda:0")
ext = tuple(float(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (10.0, 10.0, 4.0)
pts = synth.make_cloud(1_000_000, seed=0, extent=ext)
poses, quats = synth.make_path(128, optical=True, scale=ext[0] / 40.0)
cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
ws = ops.TrajWorkspace(cloud, 128)
gout = torch.ones(1, device=dev)
for _ in range(20):
    ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=0)
torch.cuda.synchronize(); print("culled ms/step %.4f" % ((time.perf_counter() - t0) * 10), ops.traj_step_stats(cloud, ws))
