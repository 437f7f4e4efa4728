"""Drop-in geometry helpers of the reference's /root/reference/src/tools.py:38-187,320-325 on MI355X:
hidden-point removal (spherical flip + convex hull), the hard frustum cull, camera intrinsics, and the
per-camera hard-visibility pipeline of /root/reference/src/pc_processor.py:158-187.

All index sets are bit-exact with the reference's CPU path (tests/test_hip_hard.py).  No CPU fallback.
"""
import types

import torch

from . import ops


def load_intrinsics(device=torch.device('cuda')):
    """/root/reference/src/tools.py:320-325 -> (K, width, height)."""
    width, height = 1232., 1616.
    K = torch.tensor([[758.03967, 0., 621.46572],
                      [0., 761.62359, 756.86402],
                      [0., 0., 1.]], dtype=torch.float32).to(device)
    return K, width, height


def sphericalFlip(points, device, param):
    """/root/reference/src/tools.py:38-53."""
    return ops.spherical_flip(torch.as_tensor(points).to(device), param)[0]


def convexHull(points, device):
    """/root/reference/src/tools.py:56-64: hull of `points` plus the origin appended as the last row.
    Returns an object with `.vertices` (ascending int32 indices, as scipy reports them in 3-D)."""
    pts = torch.as_tensor(points, dtype=torch.float32).to(device)
    idx = ops.hull_vertices_with_origin(pts)
    return types.SimpleNamespace(vertices=idx)


def hidden_pts_removal(pts: torch.Tensor, device, R_param: int = 2, duplicate_rows: str = "lowest"):
    """/root/reference/src/tools.py:67-85 -> (pts_visible (V,3), visibleMask (N,) float32 of 0/1).
    Keeps the reference's quirk: the LAST hull vertex is dropped whether or not it is the origin.

    duplicate_rows — what to do when a VISIBLE point has exact copies (identical xyz rows) elsewhere in the cloud.  The visible set
    is the same either way (count and coordinates); what differs is which of the identical rows carries the index / mask bit:
      "lowest" (default)  the copy with the lowest row index, whatever the build's schedule.  Qhull (the reference, tools.py:79)
                          reports whichever copy its insertion history met first — the first copy in ~70 % of the cases, a later one
                          otherwise — which a parallel build cannot reproduce;
      "error"             raise ValueError if that situation occurs: for callers that need Qhull's exact row indices and would
                          rather de-duplicate (a VoxelGrid-filtered cloud has no duplicates) than get a different, equally valid row."""
    if duplicate_rows not in ("lowest", "error"):
        raise ValueError('duplicate_rows must be "lowest" or "error"')
    pts = torch.as_tensor(pts).to(device)
    idx, mask = ops.hidden_pts_removal(pts, R_param)
    if duplicate_rows == "error" and idx.numel():
        _, inverse, counts = torch.unique(pts.to(torch.float32), dim=0, return_inverse=True, return_counts=True)
        dup = counts[inverse[idx.long()]] > 1
        if bool(dup.any()):
            rows = idx[dup][:8].tolist()
            raise ValueError(f"{int(dup.sum())} visible point(s) have exact duplicate rows in the cloud (e.g. rows {rows}): their indices follow the "
                             "lowest-row rule here and Qhull's insertion history in the reference; de-duplicate the cloud or pass duplicate_rows='lowest'")
    pts_visible = pts[idx.long(), :]
    return pts_visible, mask


def get_cam_frustum_pts(points, img_height, img_width, intrins, min_dist=1.0, max_dist=10.0):
    """/root/reference/src/tools.py:176-187. `points` is (3,N) in the camera frame.
    -> (points_kept (M,3), dist_mask bool (N,), fov_mask bool (N,))"""
    intr = torch.as_tensor(intrins, dtype=torch.float32)[:3, :3]
    cam = ops.Camera(intr, img_width, img_height, 1.0, 5.0)
    dist_mask, fov_mask, idx = ops.frustum_cull(points, cam, min_dist, max_dist)
    kept = points[:, idx.long()].T
    return kept, dist_mask, fov_mask


def render_pc_image(verts, K, height, width, R=None, T=None, device=torch.device('cuda'), gamma=1.0e-1, znear=1.0,
                    zfar=10.0):
    """/root/reference/src/tools.py:122-173: image (height, width, 3) of a camera-frame cloud (N,3): spheres of
    0.03 m, nearest point per pixel, white background, colours = coordinates min-max normalised over the whole
    tensor.  The reference delegates to pytorch3d's pulsar renderer, which cannot be pinned offline; this is the build's
    statement of that configuration (render_kernels.hip): every disc over a pixel weighted by its falloff and
    exp(normalised depth / gamma) — `gamma` is pulsar's blending softness, 1e-5 = the nearest point alone .. 1 = everything
    shines through; `gamma=None` gives the nearest-depth splat itself.
    R, T (pytorch3d row-vector convention X_cam = X R + T) default to the identity like the reference's."""
    v = torch.as_tensor(verts, dtype=torch.float32).to(device)
    if R is not None or T is not None:
        Rm = torch.eye(3, device=v.device) if R is None else torch.as_tensor(R, dtype=torch.float32).to(v.device).reshape(3, 3)
        Tv = torch.zeros(3, device=v.device) if T is None else torch.as_tensor(T, dtype=torch.float32).to(v.device).reshape(3)
        v = v @ Rm + Tv
    if gamma is None:
        return ops.render_points(v, K, int(height), int(width), radius=0.03, znear=znear, zfar=zfar, background=1.0)[0]
    return ops.render_points_blend(v, K, int(height), int(width), radius=0.03, znear=znear, zfar=zfar, gamma=float(gamma), background=1.0)


def zbuffer_visible_points(verts, K, height, width, znear=1.0, zfar=10.0, radius=0.03):
    """Indices of the camera-frame points that win at least one pixel of the splat: the z-buffer counterpart of
    hidden_pts_removal (resolution dependent, approximate)."""
    _, _, owns = ops.render_points(torch.as_tensor(verts), K, int(height), int(width), radius, znear, zfar)
    return torch.nonzero(owns).squeeze(1).to(torch.int32)


def denormalize(x, eps=1e-6):
    """/root/reference/src/tools.py:190-196: scale an image to 0..1 between its 2nd and 98th percentiles (for display).
    Accepts a numpy array or a tensor; returns the same kind."""
    if torch.is_tensor(x):
        flat = x.detach().to(torch.float32).flatten()
        hi, lo = torch.quantile(flat, 0.98), torch.quantile(flat, 0.02)
        return ((x - lo) / torch.clamp(hi - lo, min=eps)).clamp(0, 1)
    import numpy as np
    x_max, x_min = np.percentile(x, 98), np.percentile(x, 2)
    return ((x - x_min) / np.max([(x_max - x_min), eps])).clip(0, 1)


def ego_to_cam(points, trans, quat):
    """/root/reference/src/pc_processor.py:63-70: (N,3) ego-frame points -> (3,N) camera frame; the
    quaternion is NOT normalised there, and is not here."""
    return ops.to_camera_frame_exact(points, quat, trans, normalize=False, transpose=True)


def visible_points_from_camera(points, trans, quat, intrins, img_height, img_width, min_dist=1.0, max_dist=15.0):
    """The per-camera hard visibility pipeline of /root/reference/src/pc_processor.py:158-187 without the
    ROS glue: transform -> hard frustum cull -> HPR from the camera centre.
    -> dict(cam_points (3,N), kept_idx, kept_points (M,3), visible_idx (into kept), visible_points (V,3))"""
    cam_pts = ego_to_cam(points, trans, quat)
    intr = torch.as_tensor(intrins, dtype=torch.float32)[:3, :3]
    cam = ops.Camera(intr, img_width, img_height, 1.0, 5.0)
    _, _, kept_idx = ops.frustum_cull(cam_pts, cam, min_dist, max_dist)
    kept = cam_pts[:, kept_idx.long()].T.contiguous()
    if kept.shape[0] >= 4:
        vis_idx, _ = ops.hidden_pts_removal(kept, 2)
    else:
        vis_idx = torch.empty(0, dtype=torch.int32, device=kept.device)
    return dict(cam_points=cam_pts, kept_idx=kept_idx, kept_points=kept, visible_idx=vis_idx,
                visible_points=kept[vis_idx.long()])


def visible_points_from_cameras(points, trans, quats, intrins, img_height, img_width, min_dist=1.0, max_dist=15.0):
    """visible_points_from_camera for C cameras at once (the reference repeats the pipeline per camera topic,
    /root/reference/src/pc_processor.py:57-59,158-187): per camera transform -> hard cull, then ONE batched hull pass
    for all cameras.  trans (C,3), quats (C,4) wxyz.  -> list of dicts as visible_points_from_camera returns."""
    pts = torch.as_tensor(points, dtype=torch.float32)
    dev = pts.device
    trans = torch.as_tensor(trans, dtype=torch.float32).reshape(-1, 3).to(dev).contiguous()
    quats = torch.as_tensor(quats, dtype=torch.float32).reshape(-1, 4).to(dev).contiguous()
    intr = torch.as_tensor(intrins, dtype=torch.float32)[:3, :3]
    cam = ops.Camera(intr, img_width, img_height, 1.0, 5.0)
    C, n = trans.shape[0], pts.shape[0]
    # transform + hard cull of every camera in one batched call (quaternions NOT normalised: ego_to_cam_torch)
    kept_idx, kept_pts, counts, _ = ops.cull_waypoints(pts, trans, quats, cam, min_dist, max_dist, normalize=False)
    out = []
    for c in range(C):
        m = counts[c]
        out.append(dict(cam_points=ego_to_cam(pts, trans[c], quats[c]), kept_idx=kept_idx[c, :m], kept_points=kept_pts[c, :m]))
    offs = [0]
    for r in out:
        offs.append(offs[-1] + r["kept_points"].shape[0])
    idx, voff, _, status = ops.hidden_pts_removal_batched(torch.cat([r["kept_points"] for r in out]), offs, 2)
    for c, r in enumerate(out):
        if int(status[c]) == 3:
            raise ValueError("Points cannot contain NaN")
        if int(status[c]) == 2:
            raise RuntimeError(f"camera {c}: the culled cloud is flat, no 3-D hull (Qhull raises QH6154)")
        # fewer than 4 kept points: no hull and nothing hidden-point removal could say -> empty, as the single-camera call
        r["visible_idx"] = (idx[int(voff[c]):int(voff[c + 1])] - offs[c]).contiguous()
        r["visible_points"] = r["kept_points"][r["visible_idx"].long()]
    return out


def xy_yaw_gradient(poses_grad, quats, quats_grad):
    """(dL/dx, dL/dy, dL/dyaw) per waypoint from the gradients the models produce (`model.poses.grad`, `model.quats.grad`):
    the planar parametrisation of a ground robot's waypoint.  Yaw turns the waypoint about the world z axis,
    q(yaw) = r_z(yaw) (x) q, so dq/dyaw = 1/2 (0,0,0,1) (x) q = 1/2 (-z, -y, x, w) for q = (w, x, y, z) and
    dL/dyaw = <dL/dq, dq/dyaw>.  `quats` are the models' raw quaternions (they are normalised inside the kernels; the
    gradient w.r.t. the raw quaternion is what autograd returns).  -> (W, 3) tensor."""
    q = torch.as_tensor(quats).detach()
    gq = torch.as_tensor(quats_grad).detach()
    gp = torch.as_tensor(poses_grad).detach()
    w, x, y, z = q.unbind(-1)
    dq = 0.5 * torch.stack([-z, -y, x, w], dim=-1)
    return torch.cat([gp[..., :2], (gq * dq).sum(-1, keepdim=True)], dim=-1)
