import sys, os, time, ctypes, numpy as np, torch
sys.path.insert(0, os.getcwd())
from trajectory_optimization_amd import synth, ops, _lib
dev = torch.device("cuda:0")
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
n, w = int(os.environ.get("N", 1_000_000)), int(os.environ.get("W", 128))
modes = os.environ.get("MODES", "dense,cull").split(",")
pts = synth.make_cloud(n, seed=0)
poses, quats = synth.make_path(w, optical=True)
cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
cam = ops.Camera(K, IW, IH)
p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
ws = ops.TrajWorkspace(cloud, w)
gout = torch.ones(1, device=dev)
L = _lib.lib()
for name in modes:
    flags = ops.DENSE if name == "dense" else 0
    def step():
        half = torch.empty(cloud.n, device=dev)
        lo, mm = ops.traj_forward(cloud, p, q, cam, ws, flags=flags, rewards_half=half)
        rew, sc = ops.traj_reward(cloud, lo, cam, ws, rewards=half, prefilled=True)
        pg, qg = ops.traj_backward(cloud, w, cam, ws, lo, scalars=sc, gout=gout, flags=flags)
        return sc, pg, qg, rew
    for _ in range(5): o = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): o = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)()
    L.tohip_profile_enable(1)
    for _ in range(20): step()
    torch.cuda.synchronize()
    L.tohip_profile_read(ms, cnt)
    L.tohip_profile_enable(0)
    print(f"{n}x{w} {name}: {dt*1e3:.4f} ms/step = {n*w/dt:.3e} evals/s; kernels:", {L.tohip_profile_name(i).decode(): round(ms[i]/20*1e3, 1) for i in range(6)}, "us vis", float(o[0][1]))
