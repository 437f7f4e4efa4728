"""CPU-side checks that need no GPU: the C-ABI library loads and exports every declared symbol, argument
validation works without touching the device, the product path refuses to run without a GPU, and the
torch-side regularisers match the reference's values."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import load_golden, REPO


def test_library_exports_every_declared_symbol():
    from trajectory_optimization_amd import _lib
    header = open(os.path.join(REPO, "include", "trajopt_hip.h")).read()
    declared = set(re.findall(r"\b(tohip_\w+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    header = open(os.path.join(REPO, 'include', 'trajopt_hip.h')).read()
    assert int(re.search(r'#define TOHIP_ABI_VERSION (\d+)', header).group(1)) == _lib.ABI_VERSION == _lib.lib().tohip_abi_version()


def test_sizes_and_argument_errors_without_gpu():
    from trajectory_optimization_amd import _lib
    L = _lib.lib()
    assert L.tohip_padded_points(1) == 2048 and L.tohip_padded_points(2048) == 2048 and L.tohip_padded_points(2049) == 4096
    assert L.tohip_padded_points(0) == 0
    assert L.tohip_traj_workspace_bytes(1_000_000, 128) > 128 * 64
    assert L.tohip_traj_workspace_bytes(0, 5) == 0
    assert L.tohip_hpr_workspace_bytes(1000) > 1000 * 24
    assert L.tohip_error_string(-1).decode().startswith("invalid argument")
    # null pointers / bad sizes are rejected before any launch
    assert L.tohip_pack_cloud(None, 10, 1, None, None, 0, None) == -1
    # x|y|z, perm, bounds, inverse perm, the probe's samples
    assert L.tohip_packed_cloud_bytes(1000) == 2048 * 16 + 8 * 16 + 2048 * 4 + 8192 * 12 + 256   # ... + the header
    cam = _lib.make_camera([1, 0, 0, 0, 1, 0, 0, 0, 1], 10, 10, 1, 5)
    assert L.tohip_traj_forward(None, 10, None, None, 1, ctypes.byref(cam), None, 0, None, None, None, None, None, 0, None) == -1
    assert L.tohip_hidden_pts_removal(None, 2, 2.0, None, None, None, None, 0, None) == -1


def test_no_cpu_fallback():
    from trajectory_optimization_amd import ops
    from trajectory_optimization_amd.model import ModelTraj, ModelPose
    pts = torch.rand(100, 3)
    with pytest.raises(RuntimeError):
        ops.PackedCloud(pts)
    with pytest.raises(RuntimeError):
        ModelTraj(pts, torch.zeros(3, 3), torch.tensor([[1., 0, 0, 0]] * 3), torch.eye(3), 10., 10., device=torch.device("cpu"))
    with pytest.raises(RuntimeError):
        ModelPose(pts, torch.zeros(1, 3), torch.tensor([[1., 0, 0, 0]]), torch.eye(3), 10., 10., device=torch.device("cpu"))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under trajectory_optimization_amd/ may import, load or link it."""
    pkg = os.path.join(REPO, "trajectory_optimization_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|liboracle|oracle[/.]_build|oracle\.(traj|pose|hidden|lib|build)",
                     re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                assert not bad.search(open(os.path.join(root, f)).read()), f


def test_regularisers_match_reference():
    from trajectory_optimization_amd.model import length_calc, mean_angle_calc
    d = load_golden("funcs")
    for key in ("traj", "path27"):
        t = torch.from_numpy(d[key])
        assert abs(float(length_calc(t)) - float(d[key + "_length"])) <= 2e-6 * float(d[key + "_length"])
        assert abs(float(mean_angle_calc(t)) - float(d[key + "_mean_angle"])) <= 2e-6 * float(d[key + "_mean_angle"])
    with pytest.raises(ZeroDivisionError):
        mean_angle_calc(torch.zeros(2, 3))  # the reference divides 0.0 by (N_wps - 2) = 0


def test_sample_npz_format(tmp_path):
    """samples.load_data: both point layouts of the reference's .npz samples, identity quaternions, shape errors."""
    from trajectory_optimization_amd import samples
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(50, 3)).astype(np.float64)
    poses = rng.normal(size=(7, 3))
    np.savez(tmp_path / "a.npz", pts=pts)
    np.savez(tmp_path / "b.npz", pts=pts.T)
    np.savez(tmp_path / "p.npz", poses=poses)
    for f in ("a.npz", "b.npz"):
        x, p, q = samples.load_data(tmp_path / f, tmp_path / "p.npz")
        assert x.dtype == np.float32 and x.shape == (50, 3) and np.array_equal(x, pts.astype(np.float32))
        assert p.shape == (7, 3) and np.array_equal(q, np.tile([[1, 0, 0, 0]], (7, 1)))
    np.savez(tmp_path / "bad.npz", pts=np.zeros((4, 5)))
    with pytest.raises(ValueError):
        samples.load_data(tmp_path / "bad.npz")
    samples.save_result(tmp_path / "r.npz", poses, np.tile([[1, 0, 0, 0]], (7, 1)), rewards=np.ones(50), log={"visibility": [1.0, 1.1]})
    r = np.load(tmp_path / "r.npz")
    assert set(r.files) == {"poses", "quats_wxyz", "rewards", "log_visibility"}


def test_denormalize_like_reference():
    """tools.denormalize (tools.py:190-196): percentile stretch, numpy and torch inputs agree."""
    from trajectory_optimization_amd.tools import denormalize
    rng = np.random.default_rng(1)
    img = rng.normal(size=(40, 30, 3)).astype(np.float32) * 5 + 2
    a = denormalize(img)
    x_max, x_min = np.percentile(img, 98), np.percentile(img, 2)
    np.testing.assert_allclose(a, ((img - x_min) / max(x_max - x_min, 1e-6)).clip(0, 1), rtol=0, atol=1e-7)
    b = denormalize(torch.from_numpy(img)).numpy()
    np.testing.assert_allclose(a, b, atol=2e-6)
    assert a.min() == 0.0 and a.max() == 1.0
    assert np.all(denormalize(np.zeros((4, 4), np.float32)) == 0.0)  # flat image: eps keeps it finite


def test_committed_isa_mix_was_counted_on_these_sources():
    """bench.py prices pass 1's instruction stream from a checked-in count of the compiled loop (the latest
    profiles/rNN_pass1_isa_mix.json, tools/isa_stats.py --json): it must have been counted on the sources as they are (else the
    line says isa_mix_stale)."""
    import sys
    sys.path.insert(0, REPO)
    import bench
    mix = bench.isa_mix()
    assert not mix["stale"], f"run: python tools/isa_stats.py --json {mix['file']}  (or a new round's file)"
    cycles = 4 * mix["packed_f32"] + 8 * mix["transcendental"] + 4 * mix["other_valu"]
    assert 600 < cycles < 800 and mix["evaluations_per_iteration"] == 512
