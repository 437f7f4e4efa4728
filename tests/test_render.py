"""Row M (render_pc_image): the HIP nearest-depth splat vs its numpy restatement (parity with pulsar is unpinnable)."""
import numpy as np
import pytest
import torch



def _scene(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3)) * np.array([6.0, 6.0, 11.0]) - np.array([3.0, 3.0, 0.0])).astype(np.float32)


def test_oracle_render_basic():
    from oracle import render_oracle
    K = np.array([[50.0, 0, 32.0], [0, 50.0, 24.0], [0, 0, 1]], np.float32)
    v = np.array([[0, 0, 2.0], [0, 0, 4.0], [0.5, 0.2, 3.0], [0, 0, 0.5]], np.float32)
    img, owner = render_oracle.render_points(v, K, 48, 64, radius=0.1)
    assert owner[24, 32] == 0          # nearer of the two points on the optical axis
    assert (owner == 3).sum() == 0     # closer than znear: clipped
    assert (owner == 1).sum() == 0     # fully hidden behind point 0 (smaller disc, same centre)
    assert np.all(img[owner < 0] == 1.0) and img[owner >= 0].max() <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,seed", [(1, 48, 64, 0), (800, 96, 128, 1), (5000, 120, 90, 2)])
def test_hip_render_matches_oracle(n, h, w, seed):
    from oracle import render_oracle
    from trajectory_optimization_amd import ops
    dev = torch.device("cuda:0")
    K = np.array([[80.0, 0, w / 2], [0, 82.0, h / 2], [0, 0, 1]], np.float32)
    v = _scene(n, seed)
    img, owner, owns = ops.render_points(torch.from_numpy(v).to(dev), K, h, w, want_owner=True)
    ref_img, ref_owner = render_oracle.render_points(v, K, h, w)
    assert np.array_equal(owner.cpu().numpy(), ref_owner)
    np.testing.assert_allclose(img.cpu().numpy(), ref_img, rtol=1e-6, atol=1e-7)
    assert np.array_equal(np.flatnonzero(owns.cpu().numpy()), np.unique(ref_owner[ref_owner >= 0]))


@pytest.mark.gpu
def test_render_pc_image_api_full_size():
    """Reference call shape: render_pc_image(points (N,3) camera frame, K, height, width) at the real 1616x1232."""
    from trajectory_optimization_amd.tools import render_pc_image, zbuffer_visible_points, load_intrinsics
    dev = torch.device("cuda:0")
    K, width, height = load_intrinsics(dev)
    v = torch.from_numpy(_scene(200_000, 5)).to(dev)
    img = render_pc_image(v, K, height, width, device=dev)
    assert img.shape == (int(height), int(width), 3) and img.dtype == torch.float32
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0 and float((img != 1.0).float().mean()) > 0.01
    vis = zbuffer_visible_points(v, K, height, width)
    assert 0 < vis.numel() < v.shape[0]
    # identity extrinsics == none
    img2 = render_pc_image(v, K, height, width, R=torch.eye(3), T=torch.zeros(3), device=dev)
    assert torch.equal(img, img2)


@pytest.mark.gpu
def test_batched_zbuffer_equals_one_cloud_at_a_time():
    """tohip_zbuffer_visible_batched (all waypoints' z-buffers in the same launches: discs of near points rasterised by a whole wave,
    bids only where a plain load says they can win) marks exactly the points tohip_render_points' z-buffer lets own a pixel, cloud
    by cloud — ragged counts, an empty cloud, near points with discs of hundreds of pixels, the real image size."""
    import ctypes
    from trajectory_optimization_amd import _lib, ops
    from trajectory_optimization_amd._lib import check, ptr, stream_ptr
    from trajectory_optimization_amd.tools import load_intrinsics
    dev = torch.device("cuda:0")
    K, width, height = load_intrinsics(dev)
    counts = [60_000, 0, 17, 25_001, 3]
    n_stride = max(counts)
    verts = torch.zeros((len(counts), n_stride, 3), device=dev)
    rng = np.random.default_rng(4)
    for w, c in enumerate(counts):
        if c:
            v = (rng.random((c, 3)) * np.array([8.0, 8.0, 14.0]) - np.array([4.0, 4.0, -0.6])).astype(np.float32)   # depths 0.6 .. 14.6 m
            verts[w, :c] = torch.from_numpy(v).to(dev)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    L = _lib.lib()
    wsb = L.tohip_zbuffer_batched_workspace_bytes(int(width), int(height), len(counts))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    vis = torch.empty((len(counts), n_stride), device=dev)
    K9 = (ctypes.c_float * 9)(*K.cpu().reshape(9).tolist())
    check(L.tohip_zbuffer_visible_batched(ptr(verts), n_stride, ptr(cnt), len(counts), K9, int(width), int(height), 0.03, 1.0, 15.0, ptr(vis), ptr(ws),
                                          wsb, stream_ptr()), "tohip_zbuffer_visible_batched")
    # with room for two z-buffers only: three chunks, same result
    vis2 = torch.empty_like(vis)
    small = 2 * int(width) * int(height) * 8
    check(L.tohip_zbuffer_visible_batched(ptr(verts), n_stride, ptr(cnt), len(counts), K9, int(width), int(height), 0.03, 1.0, 15.0, ptr(vis2), ptr(ws),
                                          small, stream_ptr()), "tohip_zbuffer_visible_batched")
    assert torch.equal(vis, vis2)
    for w, c in enumerate(counts):
        assert float(vis[w, c:].abs().sum()) == 0.0
        if c == 0:
            continue
        owns = ops.render_points(verts[w, :c].contiguous(), K, height, width, znear=1.0, zfar=15.0)[2]
        assert torch.equal(vis[w, :c] != 0, owns), w
        assert 0 < int(owns.sum()) <= c
