"""Does the batched z-buffer care about the order of its points?  The same 128 culled clouds, culled in the caller's order and in the
packed (Morton) order: time of tohip_zbuffer_visible_batched alone."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import ops, synth, _lib
from trajectory_optimization_amd._lib import check, ptr, stream_ptr
W, N = 128, 1_000_000
dev = torch.device("cuda:0")
P = torch.from_numpy(synth.make_cloud(N, seed=0)).to(dev)
poses, quats = synth.make_path(W, optical=True)
poses, quats = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
cloud = ops.PackedCloud(P)
L = _lib.lib()
K9 = (ctypes.c_float * 9)(*[cam.c.K[i] for i in range(9)])
width, height = int(cam.c.img_width), int(cam.c.img_height)
wsb = L.tohip_zbuffer_batched_workspace_bytes(width, height, W)
zws = torch.empty(wsb, dtype=torch.uint8, device=dev)
visible = torch.empty(W * N, dtype=torch.float32, device=dev)
for name, rows in (("caller's order", P), ("packed order", ops._sorted_rows(cloud))):
    kept, pts_all, counts, kcnt = ops.cull_waypoints(rows, poses, quats, cam, 1.0, 15.0, normalize=True)
    def run():
        check(L.tohip_zbuffer_visible_batched(ptr(pts_all), N, ptr(kcnt), W, K9, width, height, 0.03, 1.0, 15.0, ptr(visible), ptr(zws), wsb, stream_ptr()), "zb")
    run(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    print(f"{name:16s} {1e3 * (time.perf_counter() - t) / 3:7.2f} ms   visible pairs {int(sum((visible[w * N:w * N + counts[w]] != 0).sum().item() for w in range(0, W, 16)))} (every 16th view)")
