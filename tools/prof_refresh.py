"""A few occlusion refreshes (1 M points x 128 waypoints; argv: repeats, method hpr | zbuffer) for rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import ops, synth
W, N = 128, 1_000_000
dev = torch.device("cuda:0")
P = torch.from_numpy(synth.make_cloud(N, seed=0)).to(dev)
poses, quats = synth.make_path(W, optical=True)
poses, quats = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
cloud = ops.PackedCloud(P)
method = sys.argv[2] if len(sys.argv) > 2 else "hpr"
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ops.occlusion_bits(cloud, P, poses, quats, cam, 1.0, 15.0, method)
torch.cuda.synchronize()
