"""PointCloud2 -> xyz at 1 M and 16 M points (16-byte xyz + intensity points), without and with 1 % NaN rows, ONE workspace per size carried
through the sequence dense, NaN, dense (the adaptive mode's hint flips twice; the first call after a flip runs in the "wrong" mode and must be
exact all the same): microseconds per call (HIP events around back-to-back calls through the C ABI) and GB/s on the algorithmic bytes (16 B read +
12 B written per finite point).  TOHIP_PC2_ADAPTIVE=0: always count | scan | write."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_amd import _lib, synth
from trajectory_optimization_amd._lib import ptr, stream_ptr
L = _lib.lib()
dev = torch.device("cuda:0")
for n in (1_000_000, 16_000_000):
    pts = torch.from_numpy(synth.make_cloud(n, seed=0)).to(dev)
    ws = torch.empty(L.tohip_ingest_workspace_bytes(n), dtype=torch.uint8, device=dev)
    ws.random_(0, 255)   # a fresh workspace holds anything
    for frac in (0.0, 0.01, 0.0):
        msg = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        msg[:, :3] = pts
        if frac > 0:
            g = torch.Generator(device="cpu").manual_seed(5)
            idx = torch.randperm(n, generator=g)[: int(n * frac)].to(dev)
            msg[idx, 1] = float("nan")
        raw = msg.view(torch.uint8).reshape(-1)
        out = torch.empty((n, 3), dtype=torch.float32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        fn = lambda: L.tohip_pointcloud2_to_xyz(ptr(raw), n, 16, 0, 4, 8, 7, 0, 1, ptr(out), ptr(cnt), ptr(ws), ws.numel(), stream_ptr())
        ref = msg[:, :3][torch.isfinite(msg[:, :3]).all(dim=1)]
        ok = True
        for _ in range(3):   # the first of them meets the previous message's hint
            out.fill_(-7.0)
            assert fn() == 0
            torch.cuda.synchronize()
            k = int(cnt.item())
            ok = ok and k == ref.shape[0] and torch.equal(out[:k], ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        gbs = (16.0 * n + 12.0 * k) / (us * 1e-6) / 1e9
        print(f"n={n} nan_rows={frac:.2f} kept={k} exact={ok} {us:.1f} us/call {gbs:.0f} GB/s = {gbs / 8000:.3f} of 8 TB/s", flush=True)
