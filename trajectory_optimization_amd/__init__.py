"""MI355X-native differentiable point-cloud visibility + coverage reward (the hot path of
ctu-vras/trajectory_optimization), behind the reference's ModelPose / ModelTraj / hidden_pts_removal API.

    from trajectory_optimization_amd.model import ModelPose, ModelTraj
    from trajectory_optimization_amd.tools import hidden_pts_removal, get_cam_frustum_pts, load_intrinsics

Importing this package does not touch the GPU; `model`, `tools` and `ops` load libtrajopt_hip.so on first use
and raise if it is missing (no CPU fallback).
"""
__version__ = "0.1.0"
