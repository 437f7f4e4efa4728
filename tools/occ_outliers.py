"""occlusion_bits (cull + batched HPR + bit rows) of the 1 M x 128 workload, every call timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_optimization_amd import synth, ops
dev = torch.device("cuda:0")
pts = torch.from_numpy(synth.make_cloud(1_000_000, seed=0)).to(dev)
poses, quats = synth.make_path(128, optical=True)
poses, quats = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
cloud = ops.PackedCloud(pts)
ref = None
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rows = ops.occlusion_bits(cloud, pts, poses, quats, cam, 1.0, 15.0)
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
    same = True if ref is None else torch.equal(rows, ref)
    ref = rows if ref is None else ref
    print(f"call {k}: {dt:.1f} ms same={same}", file=sys.stderr)
