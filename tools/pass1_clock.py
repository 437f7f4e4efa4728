"""In-kernel clock of k_traj_pass1 on the bench workload (diagnostic: s_memtime / s_memrealtime stamps per block)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import synth, ops, _lib
dev = torch.device("cuda:0")
n, w = int(os.environ.get('N', 1_000_000)), int(os.environ.get('W', 128))
pts = synth.make_cloud(n, seed=0)
poses, quats = synth.make_path(w, optical=True)
cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
ws = ops.TrajWorkspace(cloud, w)
L = _lib.lib()
for name, flags in (("dense", ops.DENSE),):
    nb = L.tohip_profile_clock_blocks(n, w, flags, 0)
    buf = torch.zeros(6 * nb, dtype=torch.int64, device=dev)
    t_end = time.time() + 2.0           # >= 2 s of back-to-back launches first (the guide's recipe)
    while time.time() < t_end:
        for _ in range(50):
            ops.traj_forward(cloud, p, q, cam, ws, flags=flags)
        torch.cuda.synchronize()
    L.tohip_profile_clock(buf.data_ptr())
    for _ in range(5):
        ops.traj_forward(cloud, p, q, cam, ws, flags=flags)
    torch.cuda.synchronize()
    L.tohip_profile_clock(None)
    raw = buf.cpu().numpy()
    s = raw[:2 * nb].reshape(nb, 2).astype(np.float64)
    ext = raw[2 * nb:].reshape(nb, 4)
    ok = s[:, 1] > 0
    ghz = s[ok, 0] / s[ok, 1] * 0.1
    print(f"{name}: {nb} blocks, in-kernel clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz,10):.3f}, p90 {np.percentile(ghz,90):.3f}); "
          f"block lifetime median {np.median(s[ok,1])*10:.0f} ns")
    t0 = ext[:, 0].min()
    start, end = (ext[:, 0] - t0) * 0.01, (ext[:, 1] - t0) * 0.01   # us
    hw, xcc = ext[:, 2], ext[:, 3] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    print(f"  kernel span {end.max():.1f} us; starts: p50 {np.median(start):.1f} p90 {np.percentile(start,90):.1f} max {start.max():.1f} us; distinct CU ids {len(np.unique(cu))}")
    # concurrency per CU at mid-kernel
    mid = end.max() / 2
    live = (start <= mid) & (end > mid)
    ids, counts = np.unique(cu[live], return_counts=True)
    print(f"  blocks alive at t={mid:.0f} us: {live.sum()}, per CU min/median/max {counts.min()}/{np.median(counts)}/{counts.max()} over {len(ids)} CUs")
    simd = (hw >> 4) & 3
    print("  blocks' wave-0 SIMD histogram:", np.bincount(simd.astype(int), minlength=4))
