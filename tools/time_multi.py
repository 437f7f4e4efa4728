"""optimize_trajectories (B trajectories in one pass) against B x optimize_trajectory, 1 M points x 128 waypoints each."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import ops, synth
from trajectory_optimization_amd.model import ModelTraj
from trajectory_optimization_amd.optimizer import optimize_trajectories, optimize_trajectory
dev = torch.device("cuda:0")
n, w, B, steps = int(os.environ.get("N", 1_000_000)), int(os.environ.get("W", 128)), int(os.environ.get("B", 8)), int(os.environ.get("STEPS", 20))
pts = torch.from_numpy(synth.make_cloud(n, seed=0)).to(dev)
K = torch.from_numpy(synth.K_INTRINS)
paths = []
for i in range(B):
    p, q = synth.make_path(w, optical=True)
    paths.append((torch.from_numpy(p + np.float32([0.0, 0.8 * i - 0.4 * B, 0.0])), torch.from_numpy(q)))
for dense in (False, True):
    shared = ops.PackedCloud(pts)   # one packed cloud for every model (the reference builds a model per message over the same map)

    def models():
        return [ModelTraj(shared, p, q, K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev, dense=dense) for p, q in paths]
    kw = dict(n_opt_steps=steps, lr_pose=0.02, lr_quat=0.005, rewards_th=1e9, vis_wps_dist=0.0)
    ms = models()
    optimize_trajectory(ms[0], **kw)           # warm-up
    torch.cuda.synchronize()
    ms = models()
    t0 = time.perf_counter()
    optimize_trajectory(ms[0], **kw)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / steps
    ms = models()
    t0 = time.perf_counter()
    for m in ms:
        optimize_trajectory(m, **kw)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / steps
    mb = models()
    optimize_trajectories(mb, **kw)            # warm-up (workspace allocation)
    mb = models()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    optimize_trajectories(mb, **kw)
    torch.cuda.synchronize()
    bat = (time.perf_counter() - t0) / steps
    same = all(torch.equal(a.poses.data, b.poses.data) for a, b in zip(ms, mb))
    print(f"{'dense' if dense else 'culled'}: {n} points, {B} x {w} waypoints: one trajectory {one*1e3:.3f} ms/step; {B} one after the other "
          f"{seq*1e3:.3f} ms/step; {B} in one pass {bat*1e3:.3f} ms/step = {bat/one:.2f} x one run; bitwise equal: {same}")
