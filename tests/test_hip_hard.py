"""Bit-exact parity of the hard (boolean / index) path on the GPU with the reference's CPU results."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_to_camera_frame_bit_exact(dev):
    from trajectory_optimization_amd.model import to_camera_frame
    from trajectory_optimization_amd.tools import ego_to_cam
    d = load_golden("funcs")
    cam = to_camera_frame(torch.from_numpy(d["points"]).to(dev), torch.from_numpy(d["quat"]).to(dev),
                          torch.from_numpy(d["trans"]).to(dev))
    from oracle import oracle
    assert np.array_equal(cam.cpu().numpy(), oracle.to_camera_frame(d["points"], d["quat"], d["trans"]))
    h = load_golden("hard_pipeline_bundled")
    cam3 = ego_to_cam(torch.from_numpy(h["points"]).to(dev), torch.from_numpy(h["trans"]).to(dev),
                      torch.from_numpy(h["quat"]).to(dev))
    assert np.array_equal(cam3.cpu().numpy(), h["cam"])


def test_soft_mask_functions(dev):
    from trajectory_optimization_amd.model import get_dist_mask, get_fov_mask
    d = load_golden("funcs")
    c = torch.from_numpy(d["cam"]).to(dev)
    np.testing.assert_allclose(get_dist_mask(c, 1.0, 5.0).cpu().numpy(), d["dist_mask"], rtol=2e-5, atol=1e-37)
    np.testing.assert_allclose(get_fov_mask(c, IH, IW, torch.from_numpy(K).to(dev)).cpu().numpy(), d["fov_mask"],
                               rtol=2e-5, atol=1e-37)
    fb = get_fov_mask(c, IH, IW, torch.from_numpy(K).to(dev), binary=True)
    assert np.array_equal(fb.cpu().numpy(), d["fov_mask_binary"])


@pytest.mark.parametrize("name", ["frustum_synth_10", "frustum_synth_15"])
def test_frustum_bit_exact(dev, name):
    from trajectory_optimization_amd.tools import get_cam_frustum_pts
    d = load_golden(name)
    n = d["points"].shape[0]
    cam3 = torch.from_numpy(np.ascontiguousarray(d["points"].T)).to(dev)
    kept, dist, fov = get_cam_frustum_pts(cam3, IH, IW, torch.from_numpy(K).to(dev), float(d["min_dist"]),
                                          float(d["max_dist"]))
    assert np.array_equal(dist.cpu().numpy(), np.unpackbits(d["dist_mask"])[:n].astype(bool))
    assert np.array_equal(fov.cpu().numpy(), np.unpackbits(d["fov_mask"])[:n].astype(bool))
    assert np.array_equal(kept.cpu().numpy(), d["points"][d["kept_idx"]])


def test_frustum_edge_cases(dev):
    from trajectory_optimization_amd import ops
    cam = ops.Camera(K, IW, IH)
    # nothing kept / everything kept / ragged sizes around the 1024-point tile
    for n in (1, 63, 1023, 1024, 1025, 5000):
        z = torch.full((n,), 5.0)
        pts = torch.stack([torch.zeros(n), torch.zeros(n), z]).to(dev)
        d, f, idx = ops.frustum_cull(pts, cam, 1.0, 10.0)
        assert d.all() and f.all() and torch.equal(idx.cpu(), torch.arange(n, dtype=torch.int32))
        d, f, idx = ops.frustum_cull(pts, cam, 6.0, 10.0)
        assert (~d).all() and idx.numel() == 0


def test_spherical_flip_bit_exact(dev):
    from trajectory_optimization_amd.tools import sphericalFlip
    d = load_golden("funcs")
    fl = sphericalFlip(torch.from_numpy(d["points"]), dev, 2)
    assert np.array_equal(fl.cpu().numpy(), d["flipped"])
    for name in ("hpr_synth_10k", "hpr_synth_100k", "hpr_synth_outside"):
        g = load_golden(name)
        fl = sphericalFlip(torch.from_numpy(g["points"]), dev, 2)
        assert np.array_equal(fl[:64].cpu().numpy(), g["flipped_head"])
