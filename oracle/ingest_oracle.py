"""numpy restatement of the reference's input formats (TEST INFRASTRUCTURE ONLY).

  pointcloud2_to_xyz_array   /root/reference/src/pointcloud_utils.py:22-80,180-198  (pinned: tests/golden/ingest.npz)
  pc_to_voxel                /root/reference/src/pointcloud_utils.py:279-288        (pinned: tests/golden/ingest.npz)
  voxel_grid                 pcl::VoxelGrid<PointXYZ>::applyFilter (PCL, third-party C++ absent from the reference
                             tree and from this image; restated from its published algorithm — parity unpinned):
                             used by /root/reference/launch/voxels_filtering.launch:11-21
"""
import numpy as np

_NP = {1: "i1", 2: "u1", 3: "i2", 4: "u2", 5: "i4", 6: "u4", 7: "f4", 8: "f8"}
_SZ = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 4, 8: 8}


def pointcloud2_to_xyz_array(msg, remove_nans=True):
    order = ">" if getattr(msg, "is_bigendian", False) else "<"
    dt = np.dtype({"names": [f.name for f in msg.fields],
                   "formats": [order + _NP[f.datatype] for f in msg.fields],
                   "offsets": [f.offset for f in msg.fields], "itemsize": msg.point_step})
    arr = np.frombuffer(msg.data, dtype=dt, count=msg.width * msg.height)
    if remove_nans:
        arr = arr[np.isfinite(arr["x"]) & np.isfinite(arr["y"]) & np.isfinite(arr["z"])]
    out = np.zeros((arr.shape[0], 3), dtype=np.float64)
    out[:, 0], out[:, 1], out[:, 2] = arr["x"], arr["y"], arr["z"]
    return out


def pc_to_voxel(pc, resolution=0.15, x=(0, 90), y=(-50, 50), z=(-4.5, 5.5)):
    lx = np.logical_and(pc[:, 0] >= x[0], pc[:, 0] < x[1])
    ly = np.logical_and(pc[:, 1] >= y[0], pc[:, 1] < y[1])
    lz = np.logical_and(pc[:, 2] >= z[0], pc[:, 2] < z[1])
    pc = pc[:, :3][np.logical_and(lx, np.logical_and(ly, lz))]
    pc = ((pc - np.array([x[0], y[0], z[0]])) / resolution).astype(np.int32)
    voxel = np.zeros((int((x[1] - x[0]) / resolution), int((y[1] - y[0]) / resolution), int(round((z[1] - z[0]) / resolution))))
    voxel[pc[:, 0], pc[:, 1], pc[:, 2]] = 1
    return voxel


def voxel_grid(points, leaf=0.1, field=2, lim_min=-2.5, lim_max=2.5):
    p = np.asarray(points, np.float32)
    keep = np.isfinite(p).all(1)
    if field is not None and field >= 0:
        keep &= ~((p[:, field] > np.float32(lim_max)) | (p[:, field] < np.float32(lim_min)))
    q = p[keep]
    if len(q) == 0:
        return np.zeros((0, 3), np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    cell = np.floor(q * inv).astype(np.int64)
    mn = cell.min(0)
    div = cell.max(0) - mn + 1
    if int(div[0]) * int(div[1]) * int(div[2]) > 2**31 - 1:   # pcl::VoxelGrid: "Leaf size is too small ...": output = input
        return p.copy()
    ijk = cell - mn
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    ks, qs = key[order], q[order]
    heads = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[heads[1:], len(ks)]
    out = np.empty((len(heads), 3), np.float32)
    for v, (a, b) in enumerate(zip(heads, ends)):  # sequential float32 accumulation like PCL's accumulator
        s = np.zeros(3, np.float32)
        for r in qs[a:b]:
            s = s + r
        out[v] = s / np.float32(b - a)
    return out
