"""tohip_frustum_cull at 1 M and 2 M points: microseconds per call (HIP events around back-to-back calls).  TOHIP_FRUSTUM_OWN_PREFIX=0: with the scan launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import _lib, ops, synth
from trajectory_optimization_amd._lib import ptr, stream_ptr
L = _lib.lib()
dev = torch.device("cuda:0")
cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
for n in (100_000, 1_000_000, 2_000_000):
    pts = torch.from_numpy(synth.make_cloud(n, seed=0)).to(dev)
    q, t = torch.tensor([[0.9, 0.1, -0.3, 0.2]], device=dev), torch.tensor([[6.0, 2.0, 0.0]], device=dev)
    cam3 = ops.to_camera_frame_exact(pts, q, t, normalize=True, transpose=True)
    dm, fm = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    kept, cnt = torch.empty(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(L.tohip_frustum_workspace_bytes(n), dtype=torch.uint8, device=dev)
    fn = lambda: L.tohip_frustum_cull(ptr(cam3), n, cam.ref(), 1.0, 10.0, ptr(dm), ptr(fm), ptr(kept), ptr(cnt), ptr(ws), ws.numel(), stream_ptr())
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    k = int(cnt.item())
    ref = torch.nonzero((dm != 0) & (fm != 0)).flatten().to(torch.int32)
    ok = k == ref.numel() and torch.equal(kept[:k], ref)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / 50)
    print(f"n={n} kept={k} ordered_exact={ok} {best:.2f} us/call (best of 5 x 50)", flush=True)
