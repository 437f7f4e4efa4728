import sys, os, time, ctypes, numpy as np, torch
sys.path.insert(0, os.getcwd())
from trajectory_optimization_amd import synth, ops, _lib
from oracle import oracle
dev = torch.device("cuda:0")
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT

def rel_inf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

def run(pts, poses, quats, flags, rig=None, grad_rewards=None):
    cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    rg = ops.CameraRig(rig[0], rig[1], dev) if rig is not None else None
    ws = ops.TrajWorkspace(cloud, p.shape[0] * (rg.n_cams if rg else 1))
    half = torch.empty(cloud.n, device=dev)
    lo, mm = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags, rewards_half=half)
    rew, sc = ops.traj_reward(cloud, lo, cam, ws, rewards=half, prefilled=True)
    gout = torch.ones(1, device=dev)
    pg, qg = ops.traj_backward(cloud, p.shape[0], cam, ws, lo, scalars=sc, gout=gout, rig=rg, flags=flags)
    torch.cuda.synchronize()
    return dict(lo=lo[:cloud.n].cpu().numpy(), rew=rew.cpu().numpy(), mm=mm.cpu().numpy(), sc=sc.cpu().numpy(), pg=pg.cpu().numpy(), qg=qg.cpu().numpy())

for (n, w, seed) in [(3000, 5, 1), (50_000, 16, 31), (200_000, 8, 32), (1_000_003, 3, 34)]:
    pts = synth.make_cloud(n, seed=seed)
    poses, quats = synth.make_path(w, optical=True, jitter_seed=seed)
    f = oracle.traj_forward(pts, poses, quats, K, IW, IH, prec="f64")
    pg, qg = oracle.traj_backward(pts, poses, quats, K, IW, IH, f, prec="f64")
    res = {}
    for name, flags in (("dense", ops.DENSE), ("cull", 0)):
        r = run(pts, poses, quats, flags)
        res[name] = r
        rerr = np.abs(r["rew"] - f["rewards"]) / np.abs(f["rewards"])
        print(f"n={n} w={w} {name}: vis rel {abs(r['sc'][1]-f['loss_vis'])/f['loss_vis']:.2e} rewards max rel {rerr.max():.2e} "
              f"pg {rel_inf(r['pg'], pg):.2e} qg {rel_inf(r['qg'], qg):.2e} pmax rel {np.abs(r['mm'][:,1]-f['pmax']).max()/f['pmax'].max():.1e}")
    same = all(np.array_equal(res["dense"][k], res["cull"][k], equal_nan=True) for k in ("lo", "rew", "sc", "pg", "qg"))
    print("   dense == cull bitwise:", same, " minmax equal:", np.array_equal(res["dense"]["mm"], res["cull"]["mm"]))

# timing at 1M x 128
n, w = 1_000_000, 128
pts = synth.make_cloud(n, seed=0)
poses, quats = synth.make_path(w, optical=True)
cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
cam = ops.Camera(K, IW, IH)
p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
ws = ops.TrajWorkspace(cloud, w)
gout = torch.ones(1, device=dev)
L = _lib.lib()
outs = {}
for name, flags in (("dense", ops.DENSE), ("cull", 0)):
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
    outs[name] = [t.cpu().numpy() for t in o]
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)()
    L.tohip_profile_enable(1)
    for _ in range(20): step()
    torch.cuda.synchronize()
    L.tohip_profile_read(ms, cnt)
    L.tohip_profile_enable(0)
    print(f"1Mx128 {name}: {dt*1e3:.4f} ms/step = {n*w/dt:.3e} evals/s; kernels:", {L.tohip_profile_name(i).decode(): round(ms[i]/20*1e3, 1) for i in range(6)}, "us")
print("1M dense == cull:", all(np.array_equal(a, b) for a, b in zip(outs["dense"], outs["cull"])), "vis", outs["dense"][0][1])
