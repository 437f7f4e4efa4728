"""ctypes loader of libtrajopt_hip.so (the C ABI declared in include/trajopt_hip.h).

The product path has no CPU fallback: if the HIP library is missing or does not load, importing
any op raises.  torch is imported first so that the library binds to the HIP runtime torch already
loaded (same SONAME, libamdhip64.so.7) instead of pulling a second runtime into the process.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TOHIP_LIB") or os.path.join(_HERE, "libtrajopt_hip.so")   # (TOHIP_LIB: a diagnostic build, tools/)
SRC = os.path.join(_HERE, "csrc", "trajopt_hip.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared"]

c_vp, c_i64, c_i32, c_f, c_sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_size_t


class Camera(ctypes.Structure):
    """struct tohip_camera (include/trajopt_hip.h)."""
    _fields_ = [("K", c_f * 9), ("img_width", c_f), ("img_height", c_f), ("min_dist", c_f), ("max_dist", c_f),
                ("eps", c_f)]


class Rig(ctypes.Structure):
    """struct tohip_rig (include/trajopt_hip.h)."""
    _fields_ = [("n_cams", c_i32), ("rig_quats", c_vp), ("rig_trans", c_vp)]


class TrajLoss(ctypes.Structure):
    """struct tohip_traj_loss (include/trajopt_hip.h)."""
    _fields_ = [("packed", c_vp), ("n_points", c_i64), ("n_wps", c_i64), ("wps_step", c_i32), ("flags", c_i32), ("cam", Camera),
                ("rig", Rig), ("poses0", c_vp), ("smoothness_weight", c_f), ("traj_length_weight", c_f), ("workspace", c_vp),
                ("workspace_bytes", c_sz), ("scratch", c_vp), ("scratch_bytes", c_sz), ("reg_terms", c_vp)]


class TrajOpt(ctypes.Structure):
    """struct tohip_traj_opt (include/trajopt_hip.h)."""
    _fields_ = [("packed", c_vp), ("n_points", c_i64), ("n_wps", c_i64), ("wps_step", c_i32), ("flags", c_i32), ("n_traj", c_i32),
                ("n_steps", c_i32), ("traj_offsets", c_vp), ("cam", Camera), ("rig", Rig), ("poses", c_vp), ("quats", c_vp),
                ("poses0", c_vp), ("smoothness_weight", c_f), ("traj_length_weight", c_f), ("lr_pose", c_f), ("lr_quat", c_f),
                ("beta1", c_f), ("beta2", c_f), ("adam_eps", c_f), ("rewards_th", c_f), ("smoothness_th", c_f),
                ("exp_avg_p", c_vp), ("exp_avg_sq_p", c_vp), ("exp_avg_q", c_vp), ("exp_avg_sq_q", c_vp), ("poses_grad", c_vp),
                ("quats_grad", c_vp), ("poses_grad_eval", c_vp), ("quats_grad_eval", c_vp), ("lo_sum", c_vp), ("minmax", c_vp),
                ("rewards", c_vp), ("scalars", c_vp), ("loss_log", c_vp), ("state_log", c_vp), ("workspace", c_vp),
                ("workspace_bytes", c_sz), ("scratch", c_vp), ("scratch_bytes", c_sz)]


class AdamGroup(ctypes.Structure):
    """struct tohip_adam_group (include/trajopt_hip.h)."""
    _fields_ = [("param", c_vp), ("grad", c_vp), ("exp_avg", c_vp), ("exp_avg_sq", c_vp), ("n", c_i64), ("lr", c_f), ("beta1", c_f),
                ("beta2", c_f), ("eps", c_f), ("step", c_i32)]


ADAM_MAX_GROUPS = 8  # TOHIP_ADAM_MAX_GROUPS

# name -> (restype, argtypes); every symbol include/trajopt_hip.h declares
SIGNATURES = {
    "tohip_abi_version": (ctypes.c_int, []),
    "tohip_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "tohip_padded_points": (c_i64, [c_i64]),
    "tohip_packed_cloud_bytes": (c_sz, [c_i64]),
    "tohip_pack_workspace_bytes": (c_sz, [c_i64]),
    "tohip_pack_cloud": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "tohip_traj_workspace_bytes_multi": (c_sz, [c_i64, c_i64, c_i64]),
    "tohip_traj_forward_multi": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig),
                                                 ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_reward_multi": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_f, ctypes.c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_backward_multi": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), ctypes.c_int,
                                                  c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_reward_backward_multi": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), ctypes.c_int,
                                                         c_vp, c_vp, c_f, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_reward_backward": (ctypes.c_int, [c_vp, c_i64, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), ctypes.c_int,
                                                   c_vp, c_vp, c_f, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_step_tail_multi": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp, c_vp,
                                                   c_vp, c_vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_vp, c_i64, c_vp,
                                                   c_vp]),
    "tohip_gather_waypoints_multi": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, ctypes.c_int, c_vp, c_vp, c_vp]),
    "tohip_traj_forward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig),
                                           ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_forward_backward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), ctypes.c_int,
                                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_forward_backward_multi": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(Camera),
                                                          ctypes.POINTER(Rig), ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                          c_vp, c_sz, c_vp]),
    "tohip_traj_loss_scratch_bytes": (c_sz, [c_i64, c_i64, c_i32, c_i32]),
    "tohip_traj_loss_scratch_layout": (ctypes.c_int, [c_i64, c_i64, c_i32, c_i32, ctypes.POINTER(c_i64)]),
    "tohip_traj_loss_forward": (ctypes.c_int, [ctypes.POINTER(TrajLoss), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tohip_traj_loss_backward": (ctypes.c_int, [ctypes.POINTER(TrajLoss), c_vp, c_vp, c_vp, c_vp]),
    "tohip_traj_loss_refresh": (ctypes.c_int, [ctypes.POINTER(TrajLoss), c_vp]),
    "tohip_traj_opt_scratch_bytes": (c_sz, [c_i64, c_i64]),
    "tohip_traj_opt_step": (ctypes.c_int, [ctypes.POINTER(TrajOpt), c_i32, c_vp]),
    "tohip_inverse_permutation": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "tohip_occlusion_rows": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "tohip_occlusion_rows_masked": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "tohip_zbuffer_batched_workspace_bytes": (c_sz, [c_i32, c_i32, c_i64]),
    "tohip_zbuffer_visible_batched": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(c_f), c_i32, c_i32, c_f, c_f, c_f, c_vp, c_vp,
                                                      c_sz, c_vp]),
    "tohip_occlusion_row": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tohip_traj_reward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_f, ctypes.c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_backward": (ctypes.c_int, [c_vp, c_i64, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), ctypes.c_int, c_vp,
                                            c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_pose_workspace_bytes": (c_sz, [c_i64]),
    "tohip_pose_forward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(Camera), c_vp, c_vp, c_vp, c_vp, c_sz,
                                           c_vp]),
    "tohip_pose_backward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(Camera), c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_vp, c_vp, c_sz, c_vp]),
    "tohip_pose_forward_backward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(Camera), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                    c_vp, c_sz, c_vp]),
    "tohip_pose_opt_step": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(Camera), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_vp, c_f, c_f, c_f, c_f, c_f, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "tohip_to_camera_frame": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    "tohip_soft_masks": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(Camera), c_vp, c_vp, c_vp]),
    "tohip_soft_masks_backward": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(Camera), c_vp, c_vp, c_vp, c_vp]),
    "tohip_to_camera_frame_backward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_frustum_workspace_bytes": (c_sz, [c_i64]),
    "tohip_frustum_cull": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(Camera), c_f, c_f, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_sz, c_vp]),
    "tohip_cull_waypoints_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "tohip_cull_waypoints": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.c_int, ctypes.POINTER(Camera), c_f, c_f, c_vp,
                                             c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_cull_waypoints_packed": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.c_int, ctypes.POINTER(Camera), c_f, c_f, c_vp,
                                                    c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_gather_points": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "tohip_hpr_workspace_bytes": (c_sz, [c_i64]),
    "tohip_spherical_flip": (ctypes.c_int, [c_vp, c_i64, c_f, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_hidden_pts_removal": (ctypes.c_int, [c_vp, c_i64, c_f, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_hpr_batched_workspace_bytes": (c_sz, [c_i64, ctypes.c_int32]),
    "tohip_hidden_pts_removal_batched": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, c_f, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
                                                         c_vp]),
    "tohip_convex_hull_vertices": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_vp, ctypes.POINTER(c_i32), c_vp, c_sz,
                                                   c_vp]),
    "tohip_traj_regularizers": (ctypes.c_int, [c_vp, c_vp, c_i64, c_f, c_f, c_f, c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_vp,
                                                c_vp]),
    "tohip_traj_step_tail": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                             c_vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_vp, c_vp, c_vp]),
    "tohip_gather_waypoints": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp]),
    "tohip_rows_strided": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    "tohip_adam_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f, c_f, c_f, c_f, c_i32, c_vp, c_vp]),
    "tohip_adam_step_multi": (ctypes.c_int, [ctypes.POINTER(AdamGroup), c_i32, c_vp]),
    "tohip_early_stop": (ctypes.c_int, [c_vp, c_vp, c_f, c_f, c_vp, ctypes.c_int, c_vp]),
    "tohip_ingest_workspace_bytes": (c_sz, [c_i64]),
    "tohip_pointcloud2_to_xyz": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                                 c_sz, c_vp]),
    "tohip_voxel_grid_workspace_bytes": (c_sz, [c_i64]),
    "tohip_voxel_grid": (ctypes.c_int, [c_vp, c_i64, c_f, c_f, c_f, c_i32, c_f, c_f, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_pc_to_voxel": (ctypes.c_int, [c_vp, c_i64, c_i32, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_i32, c_i32,
                                          c_i32, c_vp, c_vp]),
    "tohip_render_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "tohip_render_points": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(c_f), c_i32, c_i32, c_f, c_f, c_f, c_f, c_vp, c_vp,
                                            c_vp, c_vp, c_sz, c_vp]),
    "tohip_render_blend_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "tohip_render_points_blend": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(c_f), c_i32, c_i32, c_f, c_f, c_f, c_f, c_f, c_vp, c_vp, c_sz,
                                                  c_vp]),
    "tohip_traj_pshard_partial_count": (c_sz, [c_i64]),
    "tohip_traj_extrema_view": (ctypes.c_int, [c_i64, c_i64, c_vp, c_sz, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64)]),
    "tohip_traj_pshard_pass1": (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), ctypes.c_int,
                                                c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_pshard_local": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), ctypes.c_int, c_vp,
                                                c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_pshard_finish": (ctypes.c_int, [c_i64, c_i64, c_i64, ctypes.POINTER(Camera), ctypes.POINTER(Rig), c_vp, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_sz, c_vp]),
    "tohip_traj_candidate_flags": (ctypes.c_int, [c_i64, c_i64, c_i64, c_vp, c_sz, c_vp, c_vp]),
    "tohip_slot_flags_prefix": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "tohip_slots_pack": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.c_int, c_vp]),
    "tohip_traj_step_stats": (ctypes.c_int, [c_i64, c_i64, c_i64, c_vp, c_sz, c_vp, c_vp]),
    "tohip_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "tohip_profile_name": (ctypes.c_char_p, [ctypes.c_int]),
    "tohip_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "tohip_profile_clock": (ctypes.c_int, [c_vp]),
    "tohip_profile_clock_blocks": (c_i64, [c_i64, c_i64, ctypes.c_int, ctypes.c_int]),
    "tohip_selftest_wave_reduce": (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
}

_lib = None


def build(force=False, verbose=False):
    """Compile csrc/trajopt_hip.hip for gfx950 into libtrajopt_hip.so (in-tree)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "trajopt_hip.h"))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    cmd = [HIPCC] + HIPCC_FLAGS + [SRC, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    # into a file of this process's own, then renamed: a second process building at the same time (the ranks of a launcher) or
    # loading the library never sees half of one
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    try:
        subprocess.check_call(cmd[:-1] + [tmp])
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


def lib():
    """The loaded library with argtypes set.  Raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.tohip_abi_version() != ABI_VERSION:
            raise ImportError("libtrajopt_hip.so ABI version mismatch")
        _lib = handle
    return _lib


ABI_VERSION = 12  # TOHIP_ABI_VERSION of include/trajopt_hip.h (tests/test_host_cpu.py checks the two agree)
ENOSPC = -2  # TOHIP_ENOSPC
ENAN = -4    # TOHIP_ENAN


class HipError(RuntimeError):
    code = None


def check(code, what):
    if code == ENAN:
        raise ValueError("Points cannot contain NaN")  # what scipy.spatial.ConvexHull raises in the reference (tools.py:63)
    if code != 0:
        err = HipError(f"{what} failed: {lib().tohip_error_string(code).decode()} (code {code})")
        err.code = code
        raise err


def make_camera(K, img_width, img_height, min_dist, max_dist, eps=1e-6):
    """K: 9 floats (row-major) on the host."""
    cam = Camera()
    for i, v in enumerate(K):
        cam.K[i] = float(v)
    cam.img_width, cam.img_height = float(img_width), float(img_height)
    cam.min_dist, cam.max_dist, cam.eps = float(min_dist), float(max_dist), float(eps)
    return cam


def stream_ptr():
    return c_vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_vp(t.data_ptr()) if t is not None else c_vp(0)
