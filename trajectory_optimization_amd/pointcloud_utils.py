"""Input/output formats around the hot path, on the device: the reference's
/root/reference/src/pointcloud_utils.py (PointCloud2 <-> arrays, pc_to_voxel) and the PCL VoxelGrid nodelet its
launch files put in front of the optimisers (/root/reference/launch/voxels_filtering.launch:11-21).

No ROS here: a "message" is anything with the PointCloud2 attributes the reference reads — `data` (bytes),
`fields` (objects with name/offset/datatype), `point_step`, `width`, `height`, `is_bigendian`.
"""
import types

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

FLOAT32, FLOAT64 = 7, 8  # sensor_msgs/PointField datatype constants


def _field(msg, name):
    for f in msg.fields:
        if f.name == name:
            return f
    raise KeyError(f"PointCloud2 has no field {name!r}")


_PC2_WORKSPACES = {}


def pointcloud2_to_xyz_array(cloud_msg, remove_nans=True, device=torch.device("cuda")):
    """/root/reference/src/pointcloud_utils.py:197-198 -> (N,3) float32 tensor on `device`, message order,
    rows with a non-finite coordinate removed (the reference returns float64 and its callers cast to float32,
    /root/reference/src/trajectory_optimization.py:62-63)."""
    fx, fy, fz = (_field(cloud_msg, n) for n in "xyz")
    if not (fx.datatype == fy.datatype == fz.datatype) or fx.datatype not in (FLOAT32, FLOAT64):
        raise ValueError("x/y/z must share one of FLOAT32 / FLOAT64")
    n = int(cloud_msg.width) * int(cloud_msg.height)
    raw = np.frombuffer(bytes(cloud_msg.data), dtype=np.uint8)
    if raw.size < n * int(cloud_msg.point_step):
        raise ValueError("PointCloud2 data shorter than width*height*point_step")
    data = torch.from_numpy(raw.copy()).to(device)
    L = _lib.lib()
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=device)
    cnt = torch.zeros(1, dtype=torch.int32, device=device)
    wsb = L.tohip_ingest_workspace_bytes(n)
    # one workspace per (device, stream, size), kept between messages: it carries the library's hint whether the last message had
    # invalid rows (a dense message is then unpacked in one read, include/trajopt_hip.h)
    key = (torch.device(device).index, int(torch.cuda.current_stream(device).cuda_stream), wsb)
    ws = _PC2_WORKSPACES.get(key)
    if ws is None:
        if len(_PC2_WORKSPACES) >= 8:
            _PC2_WORKSPACES.clear()
        ws = _PC2_WORKSPACES[key] = torch.zeros(wsb, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        check(L.tohip_pointcloud2_to_xyz(ptr(data), n, int(cloud_msg.point_step), int(fx.offset), int(fy.offset),
                                         int(fz.offset), int(fx.datatype), int(bool(getattr(cloud_msg, "is_bigendian", False))),
                                         int(bool(remove_nans)), ptr(out), ptr(cnt), ptr(ws), wsb, stream_ptr()),
              "tohip_pointcloud2_to_xyz")
    return out[:int(cnt.item())]


def voxel_grid_filter(points, leaf_size=0.1, filter_field_name="z", filter_limit_min=-2.5, filter_limit_max=2.5):
    """pcl::VoxelGrid with the parameters of /root/reference/launch/voxels_filtering.launch -> (M,3) float32 centroids
    in ascending voxel-key order.  filter_field_name None disables the pass-through limits.  A grid of more than 2^31 - 1 cells is
    PCL's "leaf size is too small" case: like PCL, a warning and the input handed back."""
    pts = points.detach().to(torch.float32).contiguous()
    if not pts.is_cuda:
        raise RuntimeError("points must live on a HIP device")
    n = pts.shape[0]
    leaf = (leaf_size,) * 3 if np.isscalar(leaf_size) else tuple(leaf_size)
    field = -1 if filter_field_name is None else "xyz".index(filter_field_name)
    L = _lib.lib()
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=pts.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=pts.device)
    wsb = L.tohip_voxel_grid_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        check(L.tohip_voxel_grid(ptr(pts), n, float(leaf[0]), float(leaf[1]), float(leaf[2]), field, float(filter_limit_min),
                                 float(filter_limit_max), ptr(out), ptr(cnt), ptr(ws), wsb, stream_ptr()), "tohip_voxel_grid")
    m = int(cnt.item())
    if m < 0:   # pcl::VoxelGrid::applyFilter: the grid's cell indices would overflow an int32 -> warning, output = input
        import warnings
        warnings.warn("voxel_grid_filter: leaf size is too small for the input dataset (integer indices would overflow); returning the input")
        return pts.clone()
    return out[:m]


def pc_to_voxel(pc, resolution=0.15, x=(0, 90), y=(-50, 50), z=(-4.5, 5.5)):
    """/root/reference/src/pointcloud_utils.py:279-288 -> float64 occupancy grid tensor."""
    pts = pc.detach().to(torch.float32).contiguous()
    if not pts.is_cuda:
        raise RuntimeError("pc must live on a HIP device")
    nx, ny, nz = int((x[1] - x[0]) / resolution), int((y[1] - y[0]) / resolution), int(round((z[1] - z[0]) / resolution))
    vox = torch.empty((nx, ny, nz), dtype=torch.float64, device=pts.device)
    with torch.cuda.device(pts.device):
        check(_lib.lib().tohip_pc_to_voxel(ptr(pts), pts.shape[0], pts.shape[1], float(resolution), float(x[0]), float(x[1]),
                                           float(y[0]), float(y[1]), float(z[0]), float(z[1]), nx, ny, nz, ptr(vox),
                                           stream_ptr()), "tohip_pc_to_voxel")
    return vox


def _to_msg(points, names, stamp, frame_id):
    arr = np.asarray(points.detach().cpu().numpy() if isinstance(points, torch.Tensor) else points, np.float32)
    msg = types.SimpleNamespace()
    msg.header = types.SimpleNamespace(stamp=stamp, frame_id=frame_id)
    msg.height, msg.width = 1, arr.shape[0]
    msg.fields = [types.SimpleNamespace(name=nm, offset=4 * i, datatype=FLOAT32, count=1) for i, nm in enumerate(names)]
    msg.is_bigendian = False
    msg.point_step = 4 * len(names)
    msg.row_step = arr.shape[0]  # sic: the reference sets row_step to the point count (pointcloud_utils.py:309)
    msg.is_dense = int(np.isfinite(arr).all())
    msg.data = arr.tobytes()
    return msg


def xyz_array_to_pointcloud2(points, stamp=None, frame_id=None):
    """/root/reference/src/pointcloud_utils.py:290-313 (a plain namespace instead of a sensor_msgs object)."""
    return _to_msg(points, ("x", "y", "z"), stamp, frame_id)


def xyzi_array_to_pointcloud2(points, stamp=None, frame_id=None):
    """/root/reference/src/pointcloud_utils.py:315-338."""
    return _to_msg(points, ("x", "y", "z", "i"), stamp, frame_id)
