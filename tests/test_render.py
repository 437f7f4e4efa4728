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


def test_oracle_blend_limits():
    """The stated blend between its two ends: gamma -> 0 leaves the front sphere alone (its colour wherever its falloff is not ~0),
    a large gamma lets the hidden sphere and the background through."""
    from oracle import render_oracle
    K = np.array([[50.0, 0, 32.0], [0, 50.0, 24.0], [0, 0, 1]], np.float32)
    v = np.array([[0, 0, 2.0], [0, 0, 4.0], [0.5, 0.2, 3.0]], np.float32)
    hard, owner = render_oracle.render_points(v, K, 48, 64, radius=0.1)
    sharp = render_oracle.render_points_blend(v, K, 48, 64, radius=0.1, gamma=1e-5)
    np.testing.assert_allclose(sharp, hard, atol=1e-6)
    soft = render_oracle.render_points_blend(v, K, 48, 64, radius=0.1, gamma=1.0)
    c = (v - v.min()) / (v.max() - v.min())
    p = soft[24, 32]            # centre pixel: discs of points 0 and 1 and the background
    assert np.all(np.abs(p - c[0]) > 1e-3) and np.all(np.abs(p - 1.0) > 1e-3)
    assert np.all(soft[owner < 0] == 1.0) and soft.min() >= 0.0 and soft.max() <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,seed,gamma", [(1, 48, 64, 0, 0.1), (800, 96, 128, 1, 0.1), (5000, 120, 90, 2, 1.0), (5000, 120, 90, 2, 1e-3),
                                             (3000, 96, 128, 3, 1e-5)])
def test_hip_blend_matches_oracle(n, h, w, seed, gamma):
    """tohip_render_points_blend (fixed-point integer sums) against the float64 restatement, tolerance 1e-5 on colours in 0..1; the
    same bits on a second run; gamma = 1e-5 is the nearest-depth splat."""
    from oracle import render_oracle
    from trajectory_optimization_amd import ops
    dev = torch.device("cuda:0")
    K = np.array([[80.0, 0, w / 2], [0, 82.0, h / 2], [0, 0, 1]], np.float32)
    v = _scene(n, seed)
    vt = torch.from_numpy(v).to(dev)
    img = ops.render_points_blend(vt, K, h, w, gamma=gamma)
    ref = render_oracle.render_points_blend(v, K, h, w, gamma=gamma)
    np.testing.assert_allclose(img.cpu().numpy(), ref, rtol=0, atol=1e-5)
    assert torch.equal(img, ops.render_points_blend(vt, K, h, w, gamma=gamma))
    if gamma <= 1e-5:
        hard = ops.render_points(vt, K, h, w)[0]
        assert float((img - hard).abs().max()) < 1e-5
    from trajectory_optimization_amd import _lib
    from trajectory_optimization_amd._lib import ptr
    import ctypes
    K9 = (ctypes.c_float * 9)(*K.reshape(9).tolist())
    ws = torch.empty(_lib.lib().tohip_render_blend_workspace_bytes(w, h), dtype=torch.uint8, device=dev)
    assert _lib.lib().tohip_render_points_blend(ptr(vt), n, K9, w, h, 0.03, 1.0, 10.0, 0.0, 1.0, ptr(img), ptr(ws), ws.numel(), None) == -1   # gamma = 0: EINVAL
    assert _lib.lib().tohip_render_points_blend(ptr(vt), n, K9, w, h, 0.03, 1.0, 10.0, 0.1, 1.0, ptr(img), ptr(ws), 64, None) == -2            # ENOSPC


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
    # gamma: the default 0.1 blends, 1e-5 is the nearest-depth splat that gamma=None returns directly
    hard = render_pc_image(v, K, height, width, device=dev, gamma=None)
    diff = (render_pc_image(v, K, height, width, device=dev, gamma=1e-5) - hard).abs().amax(dim=2)
    assert float((diff > 1e-5).float().mean()) < 1e-3      # (two spheres within 1e-5 x 9 m of depth over one pixel still mix)
    assert float((img - hard).abs().max()) > 1e-2


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
