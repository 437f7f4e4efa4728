"""Stress (GPU box): PointCloud2 decoding (random point_step / field offsets / float32-float64 / NaN and inf rows), the
PCL-style voxel grid and pc_to_voxel against the numpy restatements in oracle/ingest_oracle.py.
python tools/stress_ingest.py [n_configs] [seed]"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_optimization_amd import pointcloud_utils as pcu
from oracle import ingest_oracle
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    n = int(rng.choice([0, 1, 255, 4097, 120_000]))
    f64 = bool(rng.random() < 0.4)
    size = 8 if f64 else 4
    order = rng.permutation(3)
    pad_front, pad_mid, pad_back = (int(rng.integers(0, 3)) * 4 for _ in range(3))
    offs = [0, 0, 0]
    o = pad_front
    for k, c in enumerate(order):
        offs[c] = o
        o += size + (pad_mid if k == 0 else 0)
    step = o + pad_back
    step = (step + 7) // 8 * 8 if f64 else step
    buf = rng.integers(0, 256, size=(n, step), dtype=np.uint8)
    xyz = (rng.normal(size=(n, 3)) * 20)
    if n > 3:
        bad_rows = rng.integers(0, n, max(1, n // 20))
        xyz[bad_rows, rng.integers(0, 3, len(bad_rows))] = rng.choice([np.nan, np.inf, -np.inf], len(bad_rows))
    for c in range(3):
        col = xyz[:, c].astype(np.float64 if f64 else np.float32)
        buf[:, offs[c]:offs[c] + size] = col.view(np.uint8).reshape(n, size) if n else buf[:, offs[c]:offs[c] + size]
    msg = types.SimpleNamespace(height=1, width=n, point_step=step, is_bigendian=False, data=buf.tobytes(),
                                fields=[types.SimpleNamespace(name=nm, offset=int(offs[c]), datatype=8 if f64 else 7, count=1)
                                        for c, nm in enumerate("xyz")])
    for rm in (True, False):
        got = pcu.pointcloud2_to_xyz_array(msg, remove_nans=rm, device=dev).cpu().numpy()
        ref = ingest_oracle.pointcloud2_to_xyz_array(msg, remove_nans=rm).astype(np.float32)
        if not np.array_equal(got, ref, equal_nan=True):
            bad += 1; print("pc2 mismatch", it, n, f64, step, offs, rm, got.shape, ref.shape)
    fin = xyz[np.isfinite(xyz).all(1)].astype(np.float32)
    if len(fin):
        leaf = float(rng.choice([0.1, 0.37, 1.0]))
        out = pcu.voxel_grid_filter(torch.from_numpy(xyz.astype(np.float32)).to(dev), leaf, "z", -10.0, 10.0).cpu().numpy()
        ref = ingest_oracle.voxel_grid(xyz.astype(np.float32), leaf, 2, -10.0, 10.0)
        if out.shape != ref.shape or not np.allclose(out, ref, rtol=1e-5, atol=1e-5):
            bad += 1; print("voxel grid mismatch", it, n, leaf, out.shape, ref.shape)
        res = float(rng.choice([0.25, 0.5]))
        vg = pcu.pc_to_voxel(torch.from_numpy(fin).to(dev), resolution=res, x=(-40, 40), y=(-40, 40), z=(-8, 8)).cpu().numpy()
        vr = ingest_oracle.pc_to_voxel(fin, resolution=res, x=(-40, 40), y=(-40, 40), z=(-8, 8))
        if vg.shape != vr.shape or not np.array_equal(vg > 0, vr > 0):
            bad += 1; print("pc_to_voxel mismatch", it, n, res)
print("ingest stress done, failures:", bad)
