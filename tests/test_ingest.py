"""Input formats: oracle vs the reference's golden vectors (CPU) and the HIP kernels vs both (GPU)."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden


def _msg(d, tag):
    names = "xyz"
    return types.SimpleNamespace(height=1, width=int(d["n"]), point_step=int(d[f"point_step_{tag}"]), is_bigendian=False,
                                 data=d[f"data_{tag}"].tobytes(),
                                 fields=[types.SimpleNamespace(name=c, offset=int(o), datatype=int(d[f"datatype_{tag}"]), count=1)
                                         for c, o in zip(names, d[f"offsets_{tag}"])])


@pytest.mark.parametrize("tag", ["A", "B"])
def test_oracle_pointcloud2_matches_reference(tag):
    from oracle import ingest_oracle
    d = load_golden("ingest")
    xyz = ingest_oracle.pointcloud2_to_xyz_array(_msg(d, tag))
    assert xyz.dtype == np.float64 and np.array_equal(xyz, d[f"xyz_{tag}"])


def test_oracle_pc_to_voxel_matches_reference():
    from oracle import ingest_oracle
    d = load_golden("ingest")
    vox = ingest_oracle.pc_to_voxel(d["vox_points"], resolution=0.5, x=(0, 40), y=(-20, 20), z=(-4.5, 5.5))
    assert tuple(vox.shape) == tuple(d["vox_shape"])
    assert np.array_equal(np.argwhere(vox > 0).astype(np.int32), d["vox_idx"])


def test_oracle_voxel_grid_properties():
    from oracle import ingest_oracle
    rng = np.random.default_rng(0)
    p = (rng.random((20000, 3)) * np.array([10, 10, 6]) - np.array([5, 5, 3])).astype(np.float32)
    out = ingest_oracle.voxel_grid(p, 0.5, 2, -2.5, 2.5)
    assert np.all(np.abs(out[:, 2]) <= 2.5 + 1e-6)
    cells = np.floor(out / 0.5).astype(int)
    assert len(np.unique(cells, axis=0)) == len(out)  # one centroid per voxel, each inside its voxel
    kept = p[(p[:, 2] <= 2.5) & (p[:, 2] >= -2.5)]
    assert len(out) == len(np.unique(np.floor(kept * np.float32(2.0)).astype(int), axis=0))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["A", "B"])
def test_hip_pointcloud2_to_xyz(dev, tag):
    from trajectory_optimization_amd import pointcloud_utils as pcu
    d = load_golden("ingest")
    xyz = pcu.pointcloud2_to_xyz_array(_msg(d, tag), device=dev)
    assert xyz.dtype == torch.float32
    assert np.array_equal(xyz.cpu().numpy(), d[f"xyz_{tag}"].astype(np.float32))  # == the callers' .float() cast
    raw = pcu.pointcloud2_to_xyz_array(_msg(d, tag), remove_nans=False, device=dev)
    assert raw.shape[0] == int(d["n"])


@pytest.mark.gpu
def test_hip_writer_roundtrip(dev):
    from trajectory_optimization_amd import pointcloud_utils as pcu
    d = load_golden("ingest")
    pts = torch.rand(1000, 3, device=dev) * 10 - 5
    m = pcu.xyz_array_to_pointcloud2(pts)
    assert m.point_step == int(d["writer_point_step"]) and [f.offset for f in m.fields] == [0, 4, 8]
    assert torch.equal(pcu.pointcloud2_to_xyz_array(m, device=dev), pts)
    m4 = pcu.xyzi_array_to_pointcloud2(torch.cat([pts, torch.ones(1000, 1, device=dev)], 1))
    assert m4.point_step == 16 and torch.equal(pcu.pointcloud2_to_xyz_array(m4, device=dev), pts)


@pytest.mark.gpu
def test_hip_pc_to_voxel(dev):
    from trajectory_optimization_amd import pointcloud_utils as pcu
    d = load_golden("ingest")
    vox = pcu.pc_to_voxel(torch.from_numpy(d["vox_points"]).to(dev), resolution=0.5, x=(0, 40), y=(-20, 20), z=(-4.5, 5.5))
    assert vox.dtype == torch.float64 and tuple(vox.shape) == tuple(d["vox_shape"])
    assert np.array_equal(torch.nonzero(vox > 0).cpu().numpy().astype(np.int32), d["vox_idx"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,leaf", [(1, 0.1), (5000, 0.5), (200_000, 0.1), (1_000_000, 0.2)])
def test_hip_voxel_grid_vs_oracle(dev, n, leaf):
    from oracle import ingest_oracle
    from trajectory_optimization_amd import pointcloud_utils as pcu
    rng = np.random.default_rng(n)
    p = (rng.random((n, 3)) * np.array([30, 30, 8]) - np.array([15, 15, 4])).astype(np.float32)
    if n > 10:
        p[rng.integers(0, n, n // 50)] = np.nan
    out = pcu.voxel_grid_filter(torch.from_numpy(p).to(dev), leaf, "z", -2.5, 2.5).cpu().numpy()
    ref = ingest_oracle.voxel_grid(p, leaf, 2, -2.5, 2.5) if n <= 200_000 else None
    if ref is not None:
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)
    else:  # full size: properties
        kept = p[np.isfinite(p).all(1) & (p[:, 2] <= 2.5) & (p[:, 2] >= -2.5)]
        cells = np.floor(kept * (np.float32(1.0) / np.float32(leaf))).astype(np.int64)
        assert len(out) == len(np.unique(cells, axis=0))
        np.testing.assert_allclose(out.astype(np.float64).mean(0) * 0 + np.sort(out[:, 2])[[0, -1]].clip(-2.5, 2.5).sum() * 0, 0)
    none = pcu.voxel_grid_filter(torch.from_numpy(p).to(dev), leaf, "z", 100.0, 200.0)
    assert none.shape[0] == 0


def test_oracle_voxel_grid_leaf_too_small():
    from oracle import ingest_oracle
    p = (np.random.default_rng(8).random((2000, 3)) * np.array([30, 30, 8]) - np.array([15, 15, 4])).astype(np.float32)
    assert np.array_equal(ingest_oracle.voxel_grid(p, 0.001, None), p)
    assert len(ingest_oracle.voxel_grid(p, 0.03, None)) <= 2000


@pytest.mark.gpu
def test_hip_voxel_grid_leaf_too_small_returns_the_input(dev):
    """pcl::VoxelGrid::applyFilter: when the grid's cell count exceeds an int32 ("Leaf size is too small for the input dataset.
    Integer indices would overflow.") it warns and hands back its input; so does the drop-in (the C entry point reports -1)."""
    from trajectory_optimization_amd import pointcloud_utils as pcu
    rng = np.random.default_rng(8)
    p = (rng.random((20_000, 3)) * np.array([30, 30, 8]) - np.array([15, 15, 4])).astype(np.float32)   # 30 m / 1 mm = 3e4 cells per axis: 7e12
    t = torch.from_numpy(p).to(dev)
    with pytest.warns(UserWarning, match="leaf size is too small"):
        out = pcu.voxel_grid_filter(t, 0.001, None)
    assert torch.equal(out, t) and out.data_ptr() != t.data_ptr()
    # just inside the limit it filters: 30 / 0.03 = 1000 cells per axis in x, y and 267 in z = 2.7e8
    assert 0 < pcu.voxel_grid_filter(t, 0.03, None).shape[0] <= 20_000
    # the same at the C ABI (include/trajopt_hip.h: *out_count = -1), ten times over: the count is -1 every time (no block reads the
    # count in the launch that overwrites it) and a good call on the same workspace afterwards is not disturbed by the overflow mark
    from trajectory_optimization_amd import _lib
    from trajectory_optimization_amd._lib import ptr, stream_ptr
    L = _lib.lib()
    n = t.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(L.tohip_voxel_grid_workspace_bytes(n), dtype=torch.uint8, device=dev)
    for _ in range(10):
        cnt.fill_(12345)
        assert L.tohip_voxel_grid(ptr(t), n, 0.001, 0.001, 0.001, -1, 0.0, 0.0, ptr(out), ptr(cnt), ptr(ws), ws.numel(), stream_ptr()) == 0
        assert int(cnt.item()) == -1
    assert L.tohip_voxel_grid(ptr(t), n, 0.03, 0.03, 0.03, -1, 0.0, 0.0, ptr(out), ptr(cnt), ptr(ws), ws.numel(), stream_ptr()) == 0
    assert int(cnt.item()) == pcu.voxel_grid_filter(t, 0.03, None).shape[0]

