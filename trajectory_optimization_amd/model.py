"""Drop-in ModelPose / ModelTraj (the reference's /root/reference/src/model.py) on MI355X.

Same class names, constructor arguments, attributes and forward()/criterion() semantics as the reference,
so its optimiser loops (/root/reference/src/trajectory_optimization.py:83-127,
/root/reference/src/pose_optimization.py:82-136) run unchanged: Adam over model.poses / model.quats (or
model.trans / model.quat), loss.backward(), reads of model.rewards, model.loss[...], model.observations.

What differs is inside: the visibility / reward forward and its analytic backward run as hand-written
gfx950 kernels behind the C ABI of include/trajopt_hip.h (see ops.py); the O(W) regularisers of
criterion() stay in torch.  There is no CPU fallback: constructing a model on a
non-HIP device raises.
"""
import ctypes
from copy import deepcopy
from time import time

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .optimizer import accelerate_torch_adam, tag_parameter
from .tools import hidden_pts_removal


# ------------------------------------------------------------------------------ helper functions

class _DistMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, cam):
        ctx.cam = cam
        ctx.save_for_backward(points)
        return ops.soft_masks(points, cam, want_dist=True, want_fov=False)[0]

    @staticmethod
    def backward(ctx, g):
        (points,) = ctx.saved_tensors
        return ops.soft_masks_backward(points, ctx.cam, grad_dist=g).to(points.dtype), None


class _FovMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, cam):
        ctx.cam = cam
        ctx.save_for_backward(points)
        return ops.soft_masks(points, cam, want_dist=False, want_fov=True)[1]

    @staticmethod
    def backward(ctx, g):
        (points,) = ctx.saved_tensors
        return ops.soft_masks_backward(points, ctx.cam, grad_fov=g).to(points.dtype), None


class _ToCameraFrame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, quat, trans):
        ctx.save_for_backward(verts, quat, trans)
        return ops.to_camera_frame_exact(verts, quat, trans, normalize=True)

    @staticmethod
    def backward(ctx, g):
        verts, quat, trans = ctx.saved_tensors
        gx, gq, gt = ops.to_camera_frame_backward(verts, quat, trans, g, want_points_grad=ctx.needs_input_grad[0])
        return (gx.to(verts.dtype) if gx is not None else None, gq.reshape(quat.shape).to(quat.dtype),
                gt.reshape(trans.shape).to(trans.dtype))


def get_dist_mask(points, min_dist=1.0, max_dist=5.0):
    """/root/reference/src/model.py:13-24.  HIP kernels forward and backward: differentiable w.r.t. `points` like the
    reference's torch ops."""
    assert isinstance(points, torch.Tensor)
    assert points.size()[1] == 3
    cam = ops.Camera(torch.eye(3), 1.0, 1.0, min_dist, max_dist)
    return _DistMask.apply(points, cam)


def get_fov_mask(points, img_height, img_width, intrins, eps=1e-6, binary=False):
    """/root/reference/src/model.py:27-47 (note the reference's positional order: height, then width).  The soft mask is
    differentiable w.r.t. `points`; a gradient w.r.t. the intrinsics is not provided and asking for one raises."""
    assert isinstance(points, torch.Tensor)
    assert points.size()[1] == 3
    assert isinstance(intrins, torch.Tensor)
    assert intrins.size() == torch.Size([3, 3])
    if intrins.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("get_fov_mask: no gradient w.r.t. the camera intrinsics (pass intrins.detach())")
    cam = ops.Camera(intrins, img_width, img_height, 1.0, 5.0, eps)
    if binary:
        pts3 = points.detach().to(torch.float32).t().contiguous()
        return ops.frustum_cull(pts3, cam, float("-inf"), float("inf"), want_indices=False)[1]
    return _FovMask.apply(points, cam)


def to_camera_frame(verts, quat, trans):
    """/root/reference/src/model.py:50-57, bit-identical f32 arithmetic; differentiable w.r.t. all three arguments (the
    quaternion gradient goes through F.normalize, as in the reference)."""
    assert verts.dim() == trans.dim()
    assert quat.size() == torch.Size([1, 4])
    return _ToCameraFrame.apply(verts, quat, trans)


def length_calc(traj):
    """/root/reference/src/model.py:135-139 (vectorised: one norm over the W-1 segments)."""
    if len(traj) < 2:
        return 0.0
    return torch.linalg.norm(traj[1:] - traj[:-1], dim=1).sum()


def mean_angle_calc(traj_wps, eps=1e-6):
    """/root/reference/src/model.py:142-155 (vectorised over the interior waypoints)."""
    traj_wps = torch.as_tensor(traj_wps)
    n_wps = len(traj_wps)
    if n_wps < 3:
        # the reference divides a python float 0.0 by (N_wps - 2): ZeroDivisionError for 2 waypoints
        return 0.0 / (n_wps - 2)
    ab = traj_wps[:-2] - traj_wps[1:-1]
    ac = traj_wps[2:] - traj_wps[1:-1]
    cosang = (ab * ac).sum(dim=1) / (torch.linalg.norm(ab, dim=1) * torch.linalg.norm(ac, dim=1) + eps)
    return torch.arccos(cosang).sum() / (n_wps - 2)


# ------------------------------------------------------------------------------ autograd bridges

class _PoseObservations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trans, quat, model, mask):
        t = trans.detach().contiguous()
        q = quat.detach().contiguous()
        obs, _ = ops.pose_forward(model._cloud, t, q, model._cam, model._ws, mask)
        ctx.model, ctx.mask = model, mask
        ctx.save_for_backward(t, q)
        return obs

    @staticmethod
    def backward(ctx, grad_obs):
        t, q = ctx.saved_tensors
        m = ctx.model
        tg, qg = ops.pose_backward(m._cloud, t, q, m._cam, m._ws, ctx.mask, grad_obs=grad_obs.contiguous())
        return tg, qg, None, None


class _PoseLoss(torch.autograd.Function):
    """ModelPose.forward in one autograd node: (trans, quat) -> (loss, observations).  The pass over the cloud that writes the
    observations also takes the gradient sums of the fused loss (they do not depend on its value; tohip_pose_forward_backward), so
    `loss.backward()` is a multiplication of seven numbers; a loss built on model.observations goes through the general
    dL/d observations pass."""

    @staticmethod
    def forward(ctx, trans, quat, model, mask):
        if not (trans.is_contiguous() and quat.is_contiguous() and trans.dtype == torch.float32 and quat.dtype == torch.float32):
            raise RuntimeError("ModelPose: trans / quat must be contiguous float32 tensors")
        plan = model._plan
        obs = torch.empty(plan.n, **plan.f32)
        scalars = torch.empty(4, **plan.f32)
        want_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if want_grad:
            grads = torch.empty(8, **plan.f32)   # d loss / d trans [0:3], d loss / d quat [4:8]
            plan.forward_backward(trans, quat, mask, obs, scalars, grads)
        else:
            grads = None
            plan.forward(trans, quat, mask, obs, scalars)
        ctx.model, ctx.mask, ctx.grads = model, mask, grads
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(trans, quat, scalars)   # by reference: an in-place edit before backward() raises, as for torch's ops
        loss = scalars[1]
        if model.fast_backward:
            loss = loss.as_subclass(_Loss)
        return loss, obs, scalars

    @staticmethod
    def backward(ctx, g_loss, g_obs, _g_scalars):
        t, q, scalars = ctx.saved_tensors
        m = ctx.model
        if g_loss is None and g_obs is None:
            return None, None, None, None
        if g_obs is None:
            g = g_loss.to(torch.float32) * ctx.grads
            return g[0:3].reshape(1, 3), g[4:8].reshape(1, 4), None, None
        g = g_obs.to(torch.float32)
        if g_loss is not None:
            g = g - g_loss.to(torch.float32) * scalars[1] * scalars[1]  # d loss / d observation_n = -loss^2
        tg, qg = ops.pose_backward(m._cloud, t, q, m._cam, m._ws, ctx.mask, grad_obs=g.contiguous())
        return tg, qg, None, None


def _local_forward(model, ps, qs, occ):
    """tohip_traj_forward for this rank's waypoints -> (lo_sum, rewards pre-filled with 0.5, workspace, its generation)."""
    ws = model._workspace(ps.shape[0])
    half = torch.empty(model._cloud.n, dtype=torch.float32, device=ps.device)
    lo_sum, _ = ops.traj_forward(model._cloud, ps, qs, model._cam, ws, model._rig, flags=model._flags, occ=occ, rewards_half=half)
    return lo_sum, half, ws, ws.generation


def _local_backward(model, ctx_ws, ctx_gen, ps, qs, occ, lo_sum, **kw):
    """tohip_traj_backward reads the state the step's forward left in the workspace.  If another forward has used the
    workspace since (model() called again before backward()), that state is rebuilt first — same inputs, same bits."""
    if ctx_ws.generation != ctx_gen:
        ops.traj_forward(model._cloud, ps, qs, model._cam, ctx_ws, model._rig, flags=model._flags, occ=occ)
    return ops.traj_backward(model._cloud, ps.shape[0], model._cam, ctx_ws, lo_sum, rig=model._rig, flags=model._flags, occ=occ, **kw)


class _TrajRewards(torch.autograd.Function):
    """rewards(poses, quats) for the evaluated waypoints.  With a process group the waypoints are sharded
    over the ranks; the only data-path collective is the all-reduce of the log-odds vector."""

    @staticmethod
    def forward(ctx, poses, quats, model):
        p = poses.detach().contiguous()
        q = quats.detach().contiguous()
        sh = model._shard
        lo, hi = sh.bounds(p.shape[0])
        ps, qs = p[lo:hi].clone(), q[lo:hi].clone()  # own copies: the step's inputs, whatever happens to the Parameters
        occ = None
        if hi > lo and model._occlusion is not None:
            # occlusion masks are piecewise constant in the poses: computed per forward, not differentiated
            occ = model._occlusion_rows(ps, qs)
        ws, gen, half = None, 0, None
        if hi > lo:
            lo_sum, half, ws, gen = _local_forward(model, ps, qs, occ)
        else:
            lo_sum = torch.zeros(model._cloud.npad, device=p.device)
        lo_sum = ops.allreduce_log_odds(sh, model._cloud, model._workspace(max(hi - lo, 1)), lo_sum, local=hi > lo)
        rewards, _ = ops.traj_reward(model._cloud, lo_sum, model._cam, model._workspace(max(hi - lo, 1)), rewards=half,
                                     prefilled=half is not None)
        ctx.model, ctx.range, ctx.n_wps, ctx.occ, ctx.ws, ctx.gen = model, (lo, hi), p.shape[0], occ, ws, gen
        ctx.save_for_backward(ps, qs, lo_sum)
        return rewards

    @staticmethod
    def backward(ctx, grad_rewards):
        ps, qs, lo_sum = ctx.saved_tensors
        m = ctx.model
        lo, hi = ctx.range
        # every rank fills the rows of its own waypoints; ONE (W,7) all-reduce assembles positions and quaternions
        grads = torch.zeros((ctx.n_wps, 7), dtype=torch.float32, device=lo_sum.device)
        if hi > lo:
            g = grad_rewards.to(torch.float32).contiguous()
            pg, qg = _local_backward(m, ctx.ws, ctx.gen, ps, qs, ctx.occ, lo_sum, grad_rewards=g)
            grads[lo:hi, :3], grads[lo:hi, 3:] = pg, qg
        grads = m._shard.allreduce_sum(grads)
        return grads[:, :3].contiguous(), grads[:, 3:].contiguous(), None


def _assemble_grads(model, step_w, W, lo, hi, pg, qg, g_loss, g_terms, reg_sum, reg_terms, dev):
    """(W,3) / (W,4) gradients from this rank's evaluated rows [lo, hi) (pg, qg: None when the visibility term carries no
    gradient), all-reduced over the shard, plus the regularisers' share (identical on every rank, so added after the all-reduce).
    g_terms = (g_l2, g_length, g_smooth): upstream gradients of the single entries of model.loss, or all None."""
    grads = torch.zeros((W, 7), dtype=torch.float32, device=dev)
    if pg is not None:
        rows = slice(lo * step_w, (hi - 1) * step_w + 1, step_w)
        grads[rows, :3], grads[rows, 3:] = pg, qg
    grads = model._shard.allreduce_sum(grads)
    pg_all, qg_all = grads[:, :3], grads[:, 3:]
    if all(g is None for g in g_terms):
        if g_loss is not None:
            pg_all = pg_all + g_loss * reg_sum
    else:
        for k, g in enumerate(g_terms):
            c = g_loss if g is None else (g.to(torch.float32) if g_loss is None else g_loss + g.to(torch.float32))
            if c is not None:
                pg_all = pg_all + c * reg_terms[k]
    return pg_all.contiguous(), qg_all.contiguous()


def _vis_upstream(g_loss, g_vis, g_rewards, scalars):
    """Keyword arguments of ops.traj_backward for the upstream gradients of (loss, loss['vis'], rewards); None: no gradient."""
    c_vis = g_vis if g_loss is None else (g_loss if g_vis is None else g_loss + g_vis)  # the visibility term enters the total with weight 1
    if c_vis is None and g_rewards is None:
        return None
    if g_rewards is None:
        return dict(scalars=scalars, gout=c_vis.reshape(1).contiguous())  # fused visibility loss, read on the device
    g = g_rewards.to(torch.float32)
    if c_vis is not None:
        g = g + c_vis * scalars[2]  # d loss_vis / d reward_n = -vis^2 / N
    return dict(grad_rewards=g.contiguous())


def _f32(g):
    return None if g is None else g.to(torch.float32)


class _TrajLoss(torch.autograd.Function):
    """ModelTraj.forward in one autograd node: (poses, quats) -> (loss, rewards, vis, l2, length, smooth), for a sharded and / or
    occlusion-aware model (the plain single-GPU model goes through _TrajLossPlan below: one library call per direction).

    Same launches as optimizer.optimize_trajectory's step — visibility forward, [all-reduce], reward, criterion
    regularisers with their analytic gradients — instead of the ~60 small torch kernels and as many autograd nodes
    the op-by-op criterion costs.  Every output stays differentiable, as in the reference: `loss.backward()` takes the
    fused visibility-loss path; a loss built on model.rewards or on single entries of model.loss back-propagates
    through the general dL/d rewards path and the per-term regulariser gradients."""

    @staticmethod
    def forward(ctx, poses, quats, model, step_w):
        L = _lib.lib()
        dev = poses.device
        p_all, q_all = poses.detach().contiguous(), quats.detach().contiguous()
        W = p_all.shape[0]
        p_eval = p_all[::step_w].contiguous() if step_w > 1 else p_all
        q_eval = q_all[::step_w].contiguous() if step_w > 1 else q_all
        sh = model._shard
        lo, hi = sh.bounds(p_eval.shape[0])
        ps, qs = p_eval[lo:hi].clone(), q_eval[lo:hi].clone()  # own copies: the step's inputs, whatever happens to the Parameters
        occ = None
        if hi > lo and model._occlusion is not None:
            occ = model._occlusion_rows(ps, qs)
        ws, gen, half = None, 0, None
        if hi > lo:
            lo_sum, half, ws, gen = _local_forward(model, ps, qs, occ)
        else:
            lo_sum = torch.zeros(model._cloud.npad, device=dev)
        lo_sum = ops.allreduce_log_odds(sh, model._cloud, model._workspace(max(hi - lo, 1)), lo_sum, local=hi > lo)
        rewards, scalars = ops.traj_reward(model._cloud, lo_sum, model._cam, model._workspace(max(hi - lo, 1)), rewards=half,
                                           prefilled=half is not None)
        terms = torch.empty(8, dtype=torch.float32, device=dev)
        reg_sum = torch.empty((W, 3), dtype=torch.float32, device=dev)
        reg_terms = torch.empty((3, W, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.tohip_traj_regularizers(ptr(p_all), ptr(model.poses0), W, float(model.smoothness_weight),
                                            float(model.traj_length_weight), float(model.eps), ptr(scalars), ptr(terms),
                                            ptr(reg_sum), 0, None, ptr(reg_terms), stream_ptr()), "tohip_traj_regularizers")
        ctx.model, ctx.range, ctx.step_w, ctx.W, ctx.occ, ctx.ws, ctx.gen = model, (lo, hi), step_w, W, occ, ws, gen
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(ps, qs, lo_sum, scalars, reg_sum, reg_terms)
        vis, l2, length, smooth, total = terms[:5].unbind()
        return total, rewards, vis, l2, length, smooth

    @staticmethod
    def backward(ctx, g_loss, g_rewards, g_vis, g_l2, g_length, g_smooth):
        ps, qs, lo_sum, scalars, reg_sum, reg_terms = ctx.saved_tensors
        m = ctx.model
        lo, hi = ctx.range
        g_loss = _f32(g_loss)
        pg = qg = None
        kw = _vis_upstream(g_loss, _f32(g_vis), g_rewards, scalars) if hi > lo else None
        if kw is not None:
            pg, qg = _local_backward(m, ctx.ws, ctx.gen, ps, qs, ctx.occ, lo_sum, **kw)
        pg_all, qg_all = _assemble_grads(m, ctx.step_w, ctx.W, lo, hi, pg, qg, g_loss, (g_l2, g_length, g_smooth), reg_sum, reg_terms,
                                         lo_sum.device)
        return pg_all, qg_all, None, None


class _TrajLossPoints(torch.autograd.Function):
    """ModelTraj.forward of a POINT-sharded model (distributed.PointShard): this rank's part of the cloud, every waypoint, two
    small collectives inside the forward (the waypoints' extrema after pass 1; 40 sums per waypoint and the reward sum after
    the gradient sums) — ops.PointShardStep.  The forward has the gradient of the visibility loss in hand when it returns (the
    sums are taken with unit upstream gradient), so the backward issues no kernel and no collective: it scales.  Loss, loss terms
    and gradients are identical on every rank; `rewards` are this rank's rows.  A loss built on model.rewards would need the
    other ranks' upstream gradients per point: not supported (raises)."""

    @staticmethod
    def forward(ctx, poses, quats, model, step_w):
        L = _lib.lib()
        dev = poses.device
        if not (poses.is_contiguous() and quats.is_contiguous() and poses.dtype == torch.float32 and quats.dtype == torch.float32):
            raise RuntimeError("ModelTraj: poses / quats must be contiguous float32 tensors")
        p_all, q_all = poses.detach(), quats.detach()
        W = p_all.shape[0]
        n_eval = (W + step_w - 1) // step_w
        st = model._point_step(n_eval)
        rewards, scalars, pg_e, qg_e = st.step(p_all, q_all, flags_extra=((step_w - 1) & 0xffff) << 8)   # every step_w-th waypoint, read in place
        model._mean_reward = scalars[0].clone()   # of ALL points, identical on every rank (model.mean_reward)
        terms = torch.empty(8, dtype=torch.float32, device=dev)
        reg_sum = torch.empty((W, 3), dtype=torch.float32, device=dev)
        reg_terms = torch.empty((3, W, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.tohip_traj_regularizers(ptr(p_all), ptr(model.poses0), W, float(model.smoothness_weight),
                                            float(model.traj_length_weight), float(model.eps), ptr(scalars), ptr(terms),
                                            ptr(reg_sum), 0, None, ptr(reg_terms), stream_ptr()), "tohip_traj_regularizers")
        ctx.step_w, ctx.W, ctx.n_eval = step_w, W, n_eval
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(pg_e.clone(), qg_e.clone(), reg_sum, reg_terms)
        vis, l2, length, smooth, total = terms[:5].unbind()
        return total, rewards.clone(), vis, l2, length, smooth

    @staticmethod
    def backward(ctx, g_loss, g_rewards, g_vis, g_l2, g_length, g_smooth):
        if g_rewards is not None:
            raise NotImplementedError("ModelTraj(shard=PointShard()): the rewards are rank-local; a loss built on model.rewards is not "
                                      "supported with point sharding (use WaypointShard, or model.loss / the returned loss)")
        pg_e, qg_e, reg_sum, reg_terms = ctx.saved_tensors
        g_loss, g_vis = _f32(g_loss), _f32(g_vis)
        c_vis = g_vis if g_loss is None else (g_loss if g_vis is None else g_loss + g_vis)
        dev = pg_e.device
        pg = torch.zeros((ctx.W, 3), dtype=torch.float32, device=dev)
        qg = torch.zeros((ctx.W, 4), dtype=torch.float32, device=dev)
        if c_vis is not None:
            rows = slice(0, (ctx.n_eval - 1) * ctx.step_w + 1, ctx.step_w)
            pg[rows], qg[rows] = c_vis * pg_e, c_vis * qg_e
        g_terms = (g_l2, g_length, g_smooth)
        if all(g is None for g in g_terms):
            if g_loss is not None:
                pg = pg + g_loss * reg_sum
        else:
            for k, g in enumerate(g_terms):
                c = g_loss if g is None else (g.to(torch.float32) if g_loss is None else g_loss + g.to(torch.float32))
                if c is not None:
                    pg = pg + c * reg_terms[k]
        return pg, qg, None, None


class _LossPlan:
    """A ModelTraj as the library sees it (struct tohip_traj_loss): built once per (model, waypoint step), it owns the step's
    workspace and scratch vectors, so that model() and loss.backward() are one library call each with five pointers."""

    def __init__(self, model, step_w):
        L = _lib.lib()
        dev, cloud, rig = model.device, model._cloud, model._rig
        W = model.poses.shape[0]
        C = rig.n_cams if rig is not None else 1
        self.n, self.W, self.step_w = cloud.n, W, step_w
        self.n_eval = (W + step_w - 1) // step_w
        self.ws = model._workspace(self.n_eval)
        nbytes = L.tohip_traj_loss_scratch_bytes(cloud.n, W, step_w, C)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        off = (ctypes.c_int64 * 8)()
        check(L.tohip_traj_loss_scratch_layout(cloud.n, W, step_w, C, off), "tohip_traj_loss_scratch_layout")

        def view(i, count, shape):
            return self.scratch[off[i]:off[i] + 4 * count].view(torch.float32).view(shape)
        self.poses_e, self.quats_e = view(0, 3 * self.n_eval, (self.n_eval, 3)), view(1, 4 * self.n_eval, (self.n_eval, 4))
        self.lo_sum, self.scalars = view(2, cloud.npad, (cloud.npad,)), view(4, 4, (4,))
        self.reg_sum = view(7, 3 * W, (W, 3))
        self.reg_terms = torch.empty((3, W, 3), dtype=torch.float32, device=dev)
        self.poses0 = model.poses0.contiguous()
        c = _lib.TrajLoss()
        c.packed, c.n_points, c.n_wps, c.wps_step, c.flags = cloud.blob.data_ptr(), cloud.n, W, step_w, int(model._flags)
        c.cam = model._cam.c
        if rig is not None:
            c.rig = rig.c
        c.poses0 = self.poses0.data_ptr()
        c.smoothness_weight, c.traj_length_weight = float(model.smoothness_weight), float(model.traj_length_weight)
        c.workspace, c.workspace_bytes = self.ws.buf.data_ptr(), self.ws.bytes
        c.scratch, c.scratch_bytes = self.scratch.data_ptr(), nbytes
        c.reg_terms = self.reg_terms.data_ptr()
        self.c, self.ref = c, ctypes.byref(c)
        self.model = model
        self.fwd, self.bwd, self.refresh = L.tohip_traj_loss_forward, L.tohip_traj_loss_backward, L.tohip_traj_loss_refresh
        self.sums_stale = False   # True: a general backward (tohip_traj_backward) has overwritten the unit-gradient pair sums
        self.dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.dev = dev
        self.f32 = dict(dtype=torch.float32, device=dev)
        self.one = torch.ones((), **self.f32)   # dL/d loss of a plain loss.backward()

    def forward(self, poses, quats, rewards, terms):
        idx = self.dev_index
        self.ws.generation += 1
        self.sums_stale = False
        if torch.cuda.current_device() == idx:
            rc = self.fwd(self.ref, poses.data_ptr(), quats.data_ptr(), rewards.data_ptr(), terms.data_ptr(),
                          torch._C._cuda_getCurrentRawStream(idx))
        else:
            with torch.cuda.device(idx):
                rc = self.fwd(self.ref, poses.data_ptr(), quats.data_ptr(), rewards.data_ptr(), terms.data_ptr(),
                              torch._C._cuda_getCurrentRawStream(idx))
        if rc:
            check(rc, "tohip_traj_loss_forward")
        return self.ws.generation

    def rebuild(self, poses, quats, versions):
        """The state of an earlier step, after another forward has used the workspace: possible while its inputs are unchanged."""
        if poses._version != versions[0] or quats._version != versions[1]:
            raise RuntimeError("ModelTraj: backward() of a loss whose model has been evaluated again AND whose poses / quats have been "
                               "modified by an inplace operation since: the step's state is gone (call backward() before the next "
                               "model(), or before optimizer.step())")
        return self.forward(poses, quats, torch.empty(self.n, **self.f32), torch.empty(8, **self.f32))

    def backward(self, gout, pg, qg):
        idx = self.dev_index
        with torch.cuda.device(idx) if torch.cuda.current_device() != idx else _NOOP:
            stream = torch._C._cuda_getCurrentRawStream(idx)
            if self.sums_stale:
                # a backward through model.rewards / single loss terms of this step ran before: it left ITS sums (scaled by its
                # upstream gradient) where this one expects the unit-gradient ones — take them again (same pairs, same bits)
                rc = self.refresh(self.ref, stream)
                if rc:
                    check(rc, "tohip_traj_loss_refresh")
                self.sums_stale = False
            rc = self.bwd(self.ref, gout.data_ptr(), pg.data_ptr(), qg.data_ptr(), stream)
        if rc:
            check(rc, "tohip_traj_loss_backward")


class _Noop:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOOP = _Noop()


class _FastBackward:
    """What `loss.backward()` needs when `loss` is exactly what model() returned and nothing else is asked for: the plan, the
    Parameters and their versions.  torch's autograd engine hands a HIP graph to a worker thread and waits for it (40-80 us per
    call for these two small tensors); this does what that thread would do — one library call, gradients accumulated into
    .grad — on the calling thread.  Anything beyond the plain call goes through the engine."""
    __slots__ = ("plan", "gen", "node", "params", "versions", "done")

    def __init__(self, plan, gen, node, params):
        self.plan, self.gen, self.node, self.params, self.done = plan, gen, node, params, False
        self.versions = (params[0]._version, params[1]._version)

    def usable(self, loss):
        p, q = self.params
        # (hooks registered on the Parameters' AccumulateGrad NODES — DDP does that — cannot be seen from Python: set
        # model.fast_backward = False under such wrappers)
        return (loss.grad_fn is self.node and loss._backward_hooks is None and not loss.retains_grad and
                p.requires_grad and q.requires_grad and p._backward_hooks is None and q._backward_hooks is None and
                getattr(p, "_post_accumulate_grad_hooks", None) is None and getattr(q, "_post_accumulate_grad_hooks", None) is None and
                (p.grad is None or _plain_grad(p)) and (q.grad is None or _plain_grad(q)) and not torch.is_anomaly_enabled())

    def compute(self):
        """-> the two gradients for dL/d loss = 1 (ModelTraj)."""
        plan = self.plan
        p, q = self.params
        if plan.ws.generation != self.gen:   # model() ran again since: rebuild this step's state
            self.gen = plan.rebuild(p, q, self.versions)
        pg, qg = torch.empty((plan.W, 3), **plan.f32), torch.empty((plan.W, 4), **plan.f32)
        plan.backward(plan.one, pg, qg)
        return pg, qg

    def run(self, retain_graph):
        if self.done:
            raise RuntimeError("Trying to backward through the graph a second time (or directly access saved tensors after they have "
                               "already been freed). Specify retain_graph=True if you need to backward through the graph a second time.")
        p, q = self.params
        pg, qg = self.compute()
        with torch.no_grad():
            if p.grad is None:
                p.grad = pg
            else:
                p.grad.add_(pg)
            if q.grad is None:
                q.grad = qg
            else:
                q.grad.add_(qg)
        if not retain_graph:
            self.done = True


class _FastBackwardPose(_FastBackward):
    """The same for ModelPose: the forward's pass has taken the gradient of the fused loss already.  The short cut is only taken
    while the Parameters are what the forward saw (else torch's engine raises its in-place error, as it would for torch's ops)."""
    __slots__ = ("grads",)

    def __init__(self, plan, node, params, grads):
        super().__init__(plan, 0, node, params)
        self.grads = grads

    def usable(self, loss):
        p, q = self.params
        return self.grads is not None and p._version == self.versions[0] and q._version == self.versions[1] and super().usable(loss)

    def compute(self):
        g = self.grads
        return g[0:3].reshape(1, 3).clone(), g[4:8].reshape(1, 4).clone()


class _PosePlan:
    """ModelPose's two library calls with everything constant converted once (the loop is host-bound)."""

    def __init__(self, model):
        L = _lib.lib()
        self.fwd, self.bwd, self.fwdbwd = L.tohip_pose_forward, L.tohip_pose_backward, L.tohip_pose_forward_backward
        self.blob, self.n = model._cloud.blob.data_ptr(), model._cloud.n
        self.cam = model._cam.ref()
        self.ws, self.wsb = model._ws.buf.data_ptr(), model._ws.bytes
        dev = model.device
        self.dev, self.dev_index = dev, (dev.index if dev.index is not None else torch.cuda.current_device())
        self.f32 = dict(dtype=torch.float32, device=dev)
        self.one = torch.ones(1, **self.f32)
        self.model = model

    def _call(self, fn, *args):
        idx = self.dev_index
        if torch.cuda.current_device() == idx:
            return fn(*args, self.ws, self.wsb, torch._C._cuda_getCurrentRawStream(idx))
        with torch.cuda.device(idx):
            return fn(*args, self.ws, self.wsb, torch._C._cuda_getCurrentRawStream(idx))

    def forward(self, t, q, mask, obs, scalars):
        rc = self._call(self.fwd, self.blob, self.n, t.data_ptr(), q.data_ptr(), self.cam, mask.data_ptr() if mask is not None else None,
                        obs.data_ptr(), scalars.data_ptr())
        if rc:
            check(rc, "tohip_pose_forward")

    def forward_backward(self, t, q, mask, obs, scalars, grads):
        gp = grads.data_ptr()
        rc = self._call(self.fwdbwd, self.blob, self.n, t.data_ptr(), q.data_ptr(), self.cam, mask.data_ptr() if mask is not None else None,
                        obs.data_ptr(), scalars.data_ptr(), None, gp, gp + 16)
        if rc:
            check(rc, "tohip_pose_forward_backward")

    def backward(self, t, q, mask, scalars, gout, tg, qg):
        rc = self._call(self.bwd, self.blob, self.n, t.data_ptr(), q.data_ptr(), self.cam, mask.data_ptr() if mask is not None else None, None,
                        scalars.data_ptr(), gout.data_ptr(), tg.data_ptr(), qg.data_ptr())
        if rc:
            check(rc, "tohip_pose_backward")


def _plain_grad(p):
    g = p.grad
    return g.dtype == torch.float32 and g.device == p.device and g.shape == p.shape and not g.requires_grad and not g.is_sparse


class _Loss(torch.Tensor):
    """The 0-d loss ModelTraj.forward returns: an ordinary tensor (same storage, same autograd node) whose plain
    `.backward()` skips the autograd engine's thread hand-off (_FastBackward); every other use is torch's."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        fast = self.__dict__.get("_tohip_fast")
        if fast is not None and gradient is None and inputs is None and not create_graph and fast.usable(self):
            return fast.run(retain_graph)
        return super().backward(gradient, retain_graph, create_graph, inputs)


class _TrajLossPlan(torch.autograd.Function):
    """ModelTraj.forward of a single-GPU model without occlusion rows: (poses, quats) -> (loss, rewards, vis, l2, length, smooth)
    with ONE library call in each direction (tohip_traj_loss_forward / _backward over the model's _LossPlan) and no torch kernel
    besides: the reference's `zero_grad(); loss = model(); loss.backward(); step()` loop is bound by the host on this chip.
    The backward reads the state its forward left in the plan's workspace (records, flags, regulariser gradients), not the
    Parameters: editing them in place after model() — optimizer.step() — does not disturb it.  A backward that arrives after
    ANOTHER forward of the same model re-runs its own forward first (same inputs, same bits), which needs the inputs unchanged.
    Upstream gradients other than dL/d loss (model.rewards, single entries of model.loss) take the general kernels."""

    @staticmethod
    def forward(ctx, poses, quats, plan):
        if not (poses.is_contiguous() and quats.is_contiguous() and poses.dtype == torch.float32 and quats.dtype == torch.float32):
            raise RuntimeError("ModelTraj: poses / quats must be contiguous float32 tensors")
        rewards = torch.empty(plan.n, **plan.f32)
        terms = torch.empty(8, **plan.f32)
        ctx.gen = plan.forward(poses, quats, rewards, terms)
        ctx.plan = plan
        ctx.set_materialize_grads(False)
        ctx.inputs, ctx.versions = (poses, quats), (poses._version, quats._version)
        ctx.save_for_backward(plan.one)   # nothing of it is needed: a second backward() without retain_graph raises like torch's ops
        vis, l2, length, smooth, total = terms[:5].unbind()
        if plan.model.fast_backward:
            total = total.as_subclass(_Loss)   # made here: an alias made outside would be one more autograd node
        return total, rewards, vis, l2, length, smooth

    @staticmethod
    def backward(ctx, g_loss, g_rewards, g_vis, g_l2, g_length, g_smooth):
        plan = ctx.plan
        ctx.saved_tensors
        if plan.ws.generation != ctx.gen:   # model() ran again since: rebuild this step's state (rewards / loss terms to spare vectors)
            ctx.gen = plan.rebuild(ctx.inputs[0], ctx.inputs[1], ctx.versions)
        if g_rewards is None and g_vis is None and g_l2 is None and g_length is None and g_smooth is None:
            if g_loss is None:
                return None, None, None
            if g_loss.dtype != torch.float32 or g_loss.device != plan.dev:
                g_loss = g_loss.to(**plan.f32)
            pg, qg = torch.empty((plan.W, 3), **plan.f32), torch.empty((plan.W, 4), **plan.f32)
            plan.backward(g_loss, pg, qg)
            return pg, qg, None
        m = plan.model
        g_loss = _f32(g_loss)
        pg = qg = None
        kw = _vis_upstream(g_loss, _f32(g_vis), g_rewards, plan.scalars)
        if kw is not None:
            pg, qg = ops.traj_backward(m._cloud, plan.n_eval, m._cam, plan.ws, plan.lo_sum, rig=m._rig, flags=m._flags, **kw)
            plan.sums_stale = True   # the pair sums in the workspace are now scaled by THIS upstream gradient
        pg_all, qg_all = _assemble_grads(m, plan.step_w, plan.W, 0, plan.n_eval, pg, qg, g_loss, (g_l2, g_length, g_smooth), plan.reg_sum,
                                         plan.reg_terms, plan.dev)
        return pg_all, qg_all, None


# ------------------------------------------------------------------------------ models

class ModelPose(nn.Module):
    """Single camera pose optimisation model (/root/reference/src/model.py:65-127)."""

    def __init__(self,
                 points: torch.tensor,   # the cloud, (N, 3), world frame
                 trans0: torch.tensor,   # initial camera position, shape (1, 3)
                 q0: torch.tensor,       # initial camera orientation, shape (1, 4), scalar part first
                 intrins: torch.tensor,  # pinhole matrix K, shape (3, 3)
                 img_width, img_height,
                 min_dist=1.0, max_dist=5.0,
                 device=torch.device('cuda:0'), *, fast_adam=False):
        super().__init__()
        assert trans0.size() == torch.Size([1, 3])
        assert q0.size() == torch.Size([1, 4])
        assert intrins.size() == torch.Size([3, 3])

        self.device = torch.device(device)
        # the reference keeps the caller's dtype (model.py:80) and then fails in get_fov_mask's matmul
        # for anything but float32; float32 is the contract here
        self.points = torch.as_tensor(points, dtype=torch.float32).to(self.device)
        self.rewards = None
        self.observations = None
        self.lo_sum = 0.0

        trans = torch.as_tensor(trans0, dtype=torch.float32).to(self.device)
        self.trans = nn.Parameter(trans)
        quat = torch.as_tensor(q0, dtype=torch.float32).to(self.device)
        self.quat = nn.Parameter(quat)

        self.K = torch.as_tensor(intrins, dtype=torch.float32).to(self.device)
        self.img_width, self.img_height = float(img_width), float(img_height)
        self.eps = 1e-6
        self.pc_clip_limits = [min_dist, max_dist]  # near / far range of the distance mask, metres

        self.to(self.device)
        self._cloud = ops.PackedCloud(self.points, sort=False)   # nothing culls here: the caller's order, masks and observations in place
        self._cam = ops.Camera(self.K, self.img_width, self.img_height, min_dist, max_dist, self.eps)
        self._ws = ops.PoseWorkspace(self._cloud)
        self._occlusion_mask, self._occlusion_key = None, None
        self.fused_loss = True  # forward() as one autograd node; False (or an overridden criterion): observations node + torch ops
        self.fast_backward = True   # a plain `loss.backward()` on what forward() returned runs on the calling thread
        self._plan = _PosePlan(self)
        for p in (self.trans, self.quat):
            tag_parameter(p)   # torch.optim.Adam.step() MAY update them with one launch — once the caller opts in:
        if fast_adam:          # fast_adam=True here, optimizer.accelerate_torch_adam(True) or accelerate_torch_adam(opt) (nothing is hooked otherwise)
            accelerate_torch_adam(True)

    def forward(self, debug=False, hpr=False):
        t0 = time()
        mask = None
        if hpr:
            # HPR of the WORLD-frame cloud seen from the world origin (model.py:114): pose independent,
            # so it is computed once per cloud
            key = (self.points.data_ptr(), self.points._version, tuple(self.points.shape))
            if self._occlusion_mask is None or self._occlusion_key != key:   # a replaced or edited cloud gets a new mask
                self._occlusion_mask = hidden_pts_removal(self.points.detach(), device=self.device)[1].contiguous()
                self._occlusion_key = key
            mask = self._occlusion_mask
        fused = self.fused_loss and type(self).criterion is ModelPose.criterion
        if fused:
            loss, self.observations, scalars = _PoseLoss.apply(self.trans, self.quat, self, mask)
            if type(loss) is _Loss and loss.requires_grad:
                loss.__dict__["_tohip_fast"] = _FastBackwardPose(self._plan, loss.grad_fn, (self.trans, self.quat), loss.grad_fn.grads)
        else:
            self.observations = _PoseObservations.apply(self.trans, self.quat, self, mask)
        if debug:
            torch.cuda.synchronize(self.device)
            print(f'Visibility estimation took: {1000 * (time() - t0)} msec')
            print(f'Point cloud size {self.points.size()}')
        if not fused:
            loss = self.criterion(self.observations)
        return loss

    def criterion(self, observations):
        # the more (softly) observed points, the smaller the loss: reciprocal of their sum (/root/reference/src/model.py:124-127)
        loss = 1. / (torch.sum(observations) + self.eps)
        return loss


class _NoShard:
    """Single-process placement: every waypoint is local, no collective."""
    world_size, rank = 1, 0

    @staticmethod
    def bounds(n):
        return 0, n

    @staticmethod
    def allreduce_sum(t):
        return t


class ModelTraj(nn.Module):
    """Trajectory optimisation model (/root/reference/src/model.py:158-260).

    Extra keyword arguments (absent from the reference): `rig=(quats (C,4), trans (C,3))` evaluates a rigid
    multi-camera rig at every waypoint; `shard=` a trajectory_optimization_amd.distributed.WaypointShard
    placing the waypoints over the ranks of a process group (one process per GPU, RCCL); `dense=True`
    evaluates every (point, waypoint) pair instead of skipping the pairs that provably contribute nothing;
    `occlusion='hpr'|'zbuffer'` makes the reward occlusion-aware per waypoint — the reference's TODO
    (/root/reference/src/tools.py:61-62, /root/reference/src/model.py:210): each waypoint's camera-frame cloud goes
    through the hard pipeline of /root/reference/src/pc_processor.py:171-178 (frustum cull with `occlusion_limits`,
    then HPR from the camera centre) and the points it hides get p = 0 for that waypoint.
    """

    def __init__(self,
                 points: torch.tensor,     # the cloud, (N, 3), world frame
                 wps_poses: torch.tensor,  # waypoint positions, (W, 3), one row per waypoint
                 wps_quats: torch.tensor,  # waypoint orientations, (W, 4), scalar part first
                 intrins: torch.tensor,    # pinhole matrix K, shape (3, 3)
                 img_width, img_height,
                 min_dist=1.0, max_dist=5.0,
                 smoothness_weight=14.0, traj_length_weight=0.02,
                 device=torch.device('cuda'),
                 *, rig=None, shard=None, dense=False, occlusion=None, occlusion_limits=(1.0, 15.0), occlusion_refresh_every=1,
                 occlusion_refresh_tol=None, occlusion_check_every=5, n_points_global=None, cloud=None, fast_adam=False):
        super().__init__()
        assert wps_poses.dim() == wps_quats.dim()
        assert wps_poses.size()[1] == 3
        assert wps_quats.size()[1] == 4

        self.device = torch.device(device)
        self._n_global = None
        # One packed cloud for many models: the reference builds a model per (cloud, path) message pair over the same map
        # (/root/reference/src/trajectory_optimization.py:129-136); packing — bounding box, Morton sort, tile bounds — costs several
        # optimiser steps and 20 B/point.  `points` may be an ops.PackedCloud, or `cloud=` names one (or a ModelTraj whose cloud to share).
        if isinstance(points, ops.PackedCloud):
            cloud, points = points, None
        if isinstance(cloud, ModelTraj):
            cloud = cloud._cloud
        if cloud is not None:
            if not isinstance(cloud, ops.PackedCloud) or not cloud.sorted:
                raise ValueError("cloud= must be an ops.PackedCloud in Morton order (sort=True) or a ModelTraj")
            if shard is not None and getattr(shard, "kind", "waypoints") == "points":
                raise ValueError("a shared packed cloud holds the whole cloud: not available with PointShard")
            if cloud.device != (self.device if self.device.index is not None else torch.device(self.device.type, torch.cuda.current_device())):
                raise ValueError(f"the packed cloud lives on {cloud.device}, the model on {self.device}")
            if points is not None and not (torch.is_tensor(points) and points.data_ptr() == cloud.points.data_ptr() and
                                           tuple(points.shape) == tuple(cloud.points.shape) and points.stride() == cloud.points.stride()):
                # a different tensor object: it must hold the packed cloud's rows (an equal-sized OTHER cloud would silently be
                # replaced by cloud.points otherwise); the comparison is one pass over the rows, paid only by callers who hand both
                pt = torch.as_tensor(points, dtype=torch.float32)
                if tuple(pt.shape) != tuple(cloud.points.shape) or not torch.equal(pt.to(cloud.points.device), cloud.points):
                    raise ValueError("cloud= does not hold these points")
            self.points = cloud.points
        elif shard is not None and getattr(shard, "kind", "waypoints") == "points":
            # point sharding: this rank keeps its own rows of the cloud (the whole cloud is handed in, or — n_points_global — the
            # rows already); model.rewards are those rows' rewards
            pts = torch.as_tensor(points, dtype=torch.float32)
            if n_points_global is None:
                self._n_global = int(pts.shape[0])
                lo_p, hi_p = shard.point_bounds(self._n_global)
                pts = pts[lo_p:hi_p]
            else:
                self._n_global = int(n_points_global)
            if occlusion is not None:
                raise ValueError("occlusion-aware rewards need the whole cloud on every rank: not available with PointShard")
            self.points = pts.contiguous().to(self.device)
        else:
            self.points = torch.as_tensor(points, dtype=torch.float32).to(self.device)
        self.rewards = None
        self._mean_reward = None
        self.observations = None
        self.lo_sum = 0.0  # attribute kept for the reference's surface (its accumulated log-odds); the kernels hold theirs in packed order

        self.poses0 = torch.as_tensor(wps_poses, dtype=torch.float32).to(self.device)  # the trajectory as given: criterion measures against it
        self.quats0 = torch.as_tensor(wps_quats, dtype=torch.float32).to(self.device)

        self.poses = nn.Parameter(deepcopy(self.poses0))
        self.quats = nn.Parameter(deepcopy(self.quats0))

        self.K = torch.as_tensor(intrins, dtype=torch.float32).to(self.device)
        self.img_width, self.img_height = float(img_width), float(img_height)
        self.eps = 1e-6
        self.pc_clip_limits = [min_dist, max_dist]  # near / far range of the distance mask, metres

        self.loss = {'vis': float('inf'),
                     'length': float('inf'),
                     'l2': float('inf'),
                     'smooth': float('inf')}
        self.smoothness_weight = smoothness_weight
        self.traj_length_weight = traj_length_weight

        self.to(self.device)
        self._cloud = cloud if cloud is not None else ops.PackedCloud(self.points)
        self._cam = ops.Camera(self.K, self.img_width, self.img_height, min_dist, max_dist, self.eps)
        self._rig = ops.CameraRig(rig[0], rig[1], self.device) if rig is not None else None
        self._shard = shard if shard is not None else _NoShard()
        self._flags = ops.DENSE if dense else 0  # dense: evaluate every pair (results are bitwise the same)
        if occlusion not in (None, "hpr", "zbuffer"):
            raise ValueError("occlusion must be None, 'hpr' or 'zbuffer'")
        self._occlusion, self._occlusion_limits = occlusion, occlusion_limits
        # The occlusion masks are piecewise constant in the poses (a point is hidden from a waypoint or it is not) and carry no
        # gradient; building them — a hard cull and a convex hull (or a z-buffer) per waypoint — costs hundreds of plain steps.
        # occlusion_refresh_every = k: they are rebuilt on every k-th forward of the model (k = 1: every forward, the bits of a
        # model without the policy) and reused in between; refresh_occlusion() forces a rebuild at the next forward.
        # occlusion_refresh_tol = metres, or (metres, radians): the MOTION-triggered policy — every occlusion_check_every-th
        # forward looks (one small device-to-host read) whether a waypoint has moved or turned by more than that since ITS rows were
        # built; if one has, the rows of every waypoint beyond half the tolerance are rebuilt (one batched pass for them) and the
        # others kept: a waypoint that has converged stops paying.  occlusion_refresh_every stays the cap on a row's age.
        self.occlusion_refresh_every = max(1, int(occlusion_refresh_every))
        self._occ_cache = None   # (rows, shape key, forwards since the last FULL rebuild)
        self._occ_built = None   # the body poses each waypoint's rows were built for: (positions, normalised quaternions)
        self.occlusion_refresh_tol = occlusion_refresh_tol   # (property: a scalar becomes (metres, radians); setting it later restarts the policy)
        self.occlusion_check_every = max(1, int(occlusion_check_every))
        self.occlusion_rebuilds = [0, 0]   # (full rebuilds, waypoints rebuilt by the motion policy): bookkeeping for tools and tests
        self._ws_cache = {}
        self._plan_obj, self._plan_key = None, None
        self._wps_step_cache = {}
        self._length0 = None
        # forward() as ONE autograd node (visibility + criterion on the device); False: rewards node + the op-by-op
        # torch criterion below (always used when a subclass overrides criterion, or for fewer than 3 waypoints)
        self.fused_loss = True
        # a plain `loss.backward()` on what forward() returned runs on the calling thread (_FastBackward); False: always torch's engine
        self.fast_backward = True
        for p in (self.poses, self.quats):
            tag_parameter(p)   # torch.optim.Adam.step() MAY update them with one launch — once the caller opts in:
        if fast_adam:          # fast_adam=True here, optimizer.accelerate_torch_adam(True) or accelerate_torch_adam(opt) (nothing is hooked otherwise)
            accelerate_torch_adam(True)

    @classmethod
    def sharing_cloud_of(cls, other, wps_poses, wps_quats, **kw):
        """A model of another trajectory over `other`'s cloud: same packed cloud (not packed again), camera and device; keyword
        arguments as the constructor's (weights, rig, dense, ...)."""
        kw.setdefault("device", other.device)
        kw.setdefault("min_dist", other.pc_clip_limits[0])
        kw.setdefault("max_dist", other.pc_clip_limits[1])
        return cls(other._cloud, wps_poses, wps_quats, other.K, other.img_width, other.img_height, **kw)

    @property
    def mean_reward(self):
        """mean(rewards) over the WHOLE cloud after the last forward, as a 0-d tensor — what the reference's early-stop rule reads
        (`torch.mean(model.rewards) / reward0`, /root/reference/src/trajectory_optimization.py:119-122).  With point sharding
        model.rewards holds this rank's rows only and torch.mean of it differs from rank to rank: a loop that stops on it would
        leave the ranks at different steps, and the ones that go on would wait for ever in the next forward's collectives.  This is
        the replicated value (the all-reduced reward sum over the global point count); without point sharding it is
        torch.mean(model.rewards)."""
        if self._n_global is not None:
            return self._mean_reward
        return torch.mean(self.rewards.detach()) if self.rewards is not None else None

    def refresh_occlusion(self):
        """The next forward rebuilds the occlusion masks whatever occlusion_refresh_every says."""
        self._occ_cache = None

    @property
    def occlusion_refresh_tol(self):
        return self._occ_tol

    @occlusion_refresh_tol.setter
    def occlusion_refresh_tol(self, tol):
        """None, metres, or (metres, radians).  Public like occlusion_refresh_every: changing it after a forward drops the cached rows,
        so the next forward is a full rebuild that records the poses the motion policy compares against."""
        if tol is not None and not isinstance(tol, (tuple, list)):
            tol = (float(tol), 0.35 * float(tol))   # (1 rad turns a point 3 m away by 3 m)
        self._occ_tol = tuple(float(x) for x in tol) if tol is not None else None
        self._occ_cache, self._occ_built = None, None

    def _occlusion_rows(self, ps, qs):
        """Occlusion bit rows of the given body waypoints, one row per virtual waypoint v = w*C + c (with a rig: the cameras'
        own poses t_v = t_w + R(q_w) l_c, q_v = q_w/|q_w| (x) q_c — the composition the kernels apply).  Rebuilt as a whole on
        every occlusion_refresh_every-th call; with occlusion_refresh_tol set, in between, the rows of the waypoints that have
        moved (see the constructor); reused otherwise."""
        key = (tuple(ps.shape), tuple(qs.shape))
        c = self._occ_cache
        if c is None or c[1] != key or c[2] >= self.occlusion_refresh_every or (self.occlusion_refresh_tol is not None and self._occ_built is None):
            rows = self._build_occlusion_rows(ps, qs)
            self._occ_cache = (rows, key, 1)
            self.occlusion_rebuilds[0] += 1
            if self.occlusion_refresh_tol is not None:
                self._occ_built = (ps.clone(), torch.nn.functional.normalize(qs, dim=1))
            return rows
        rows, age = c[0], c[2]
        if self.occlusion_refresh_tol is not None and age % self.occlusion_check_every == 0:
            tol_p, tol_q = self.occlusion_refresh_tol
            bp, bq = self._occ_built
            qn = torch.nn.functional.normalize(qs, dim=1)
            dist = (ps - bp).norm(dim=1)
            ang = 2.0 * torch.arccos((qn * bq).sum(dim=1).abs().clamp(max=1.0))   # the rotation between the two orientations
            if bool(((dist > tol_p) | (ang > tol_q)).any()):   # (the policy's host read)
                idx = ((dist > 0.5 * tol_p) | (ang > 0.5 * tol_q)).nonzero().flatten()
                C = self._rig.n_cams if self._rig is not None else 1
                new = self._build_occlusion_rows(ps[idx].contiguous(), qs[idx].contiguous())
                vidx = (idx[:, None] * C + torch.arange(C, device=idx.device)[None, :]).flatten()
                rows = rows.clone()   # (a step whose backward is still to come holds the old rows)
                rows[vidx] = new
                bp[idx], bq[idx] = ps[idx], qn[idx]
                self.occlusion_rebuilds[1] += int(idx.numel())
        self._occ_cache = (rows, key, age + 1)
        return rows

    def _build_occlusion_rows(self, ps, qs):
        if self._rig is not None:
            qn = qs / qs.norm(dim=1, keepdim=True).clamp_min(1e-12)
            qc, lc = self._rig.q, self._rig.t
            aw, ax, ay, az = qn[:, None, :].unbind(-1)
            bw, bx, by, bz = qc[None, :, :].unbind(-1)
            vq = torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                              aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1).reshape(-1, 4)
            w, x, y, z = qn.unbind(-1)
            R = torch.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
                             2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                             2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z], -1).reshape(-1, 3, 3)
            vt = (ps[:, None, :] + torch.einsum("wij,cj->wci", R, lc)).reshape(-1, 3)
            ps, qs = vt.contiguous(), vq.contiguous()
        return ops.occlusion_bits(self._cloud, self.points, ps, qs, self._cam, self._occlusion_limits[0],
                                  self._occlusion_limits[1], self._occlusion)

    def _workspace(self, n_local_wps):
        v = n_local_wps * (self._rig.n_cams if self._rig is not None else 1)
        ws = self._ws_cache.get(v)
        if ws is None:
            ws = self._ws_cache[v] = ops.TrajWorkspace(self._cloud, v)
        return ws

    def _point_step(self, n_eval):
        """The point-sharded step's buffers (ops.PointShardStep) for n_eval evaluated waypoints."""
        st = self._ws_cache.get(("points", n_eval))
        if st is None:
            st = self._ws_cache[("points", n_eval)] = ops.PointShardStep(self._cloud, self._n_global, n_eval, self._cam, self._workspace(n_eval),
                                                                         self._shard, rig=self._rig, flags=self._flags)
        return st

    def _plan(self, step_w):
        """The library-side description of this model for the one-call forward / backward (rebuilt when something it froze
        has changed: the weights of criterion, the mode, the initial trajectory, the number of waypoints)."""
        key = (step_w, float(self.smoothness_weight), float(self.traj_length_weight), self._flags, self.poses0.data_ptr(),
               self.poses.shape[0])
        if self._plan_key != key:
            self._plan_obj, self._plan_key = _LossPlan(self, step_w), key
        return self._plan_obj

    def _wps_step(self, vis_wps_dist):
        # based on the mean waypoint distance of the INITIAL trajectory (model.py:214-215); constant per
        # model, so the host sync the reference pays on every forward happens once
        step = self._wps_step_cache.get(vis_wps_dist)
        if step is None:
            mean_wps_dist = (self.poses0[1:, :] - self.poses0[:-1, :]).norm(dim=1).mean()
            step = int(vis_wps_dist / mean_wps_dist) + 1
            self._wps_step_cache[vis_wps_dist] = step
        return step

    def forward(self,
                vis_wps_dist=0.5,  # metres between the waypoints whose visibility is evaluated (every wps_step-th one)
                debug=False):
        """/root/reference/src/model.py:200-242: soft visibility of every point from every wps_step-th waypoint, normalised per
        waypoint, clipped, turned into log-odds and summed over the waypoints; rewards = sigmoid of the sum; returns criterion()."""
        t0 = time()
        N_wps = len(self.poses)
        wps_step = self._wps_step(vis_wps_dist)
        if self._n_global is not None:
            if not (N_wps >= 3 and type(self).criterion is ModelTraj.criterion):
                raise NotImplementedError("ModelTraj(shard=PointShard()) supports the reference's criterion on >= 3 waypoints")
            loss, self.rewards, vis, l2, length, smooth = _TrajLossPoints.apply(self.poses, self.quats, self, wps_step)
            self.loss = {'vis': vis, 'length': length, 'l2': l2, 'smooth': smooth}
            return loss
        if self.fused_loss and N_wps >= 3 and type(self).criterion is ModelTraj.criterion:
            if self._occlusion is None and self._shard.world_size == 1 and not getattr(self._shard, "_always", False):
                plan = self._plan(wps_step)
                loss, self.rewards, vis, l2, length, smooth = _TrajLossPlan.apply(self.poses, self.quats, plan)
                if type(loss) is _Loss and loss.requires_grad:
                    loss.__dict__["_tohip_fast"] = _FastBackward(plan, plan.ws.generation, loss.grad_fn, (self.poses, self.quats))
            else:
                loss, self.rewards, vis, l2, length, smooth = _TrajLoss.apply(self.poses, self.quats, self, wps_step)
            self.loss = {'vis': vis, 'length': length, 'l2': l2, 'smooth': smooth}
            if debug:
                torch.cuda.synchronize(self.device)
                print(f'Trajectory evaluation took {1000 * (time() - t0)} msec')
            return loss
        if wps_step == 1:
            poses_eval, quats_eval = self.poses, self.quats
        else:
            idx = torch.arange(0, N_wps, wps_step, device=self.device)
            poses_eval, quats_eval = self.poses.index_select(0, idx), self.quats.index_select(0, idx)
        self.rewards = _TrajRewards.apply(poses_eval, quats_eval, self)  # total trajectory observations
        if debug:
            torch.cuda.synchronize(self.device)
            print(f'Trajectory evaluation took {1000 * (time() - t0)} msec')
        loss = self.criterion(self.rewards)
        return loss

    def criterion(self, rewards):
        # the four terms of /root/reference/src/model.py:244-260, op by op (the fused node computes the same on the device)
        # visibility: reciprocal of the mean reward
        self.loss['vis'] = 1. / (torch.mean(rewards) + self.eps)

        # the first waypoint should stay where the trajectory started
        self.loss['l2'] = torch.linalg.norm(self.poses[0] - self.poses0[0])

        # straighter is better: weight over the mean interior angle of the polyline
        self.loss['smooth'] = self.smoothness_weight / (mean_angle_calc(self.poses, self.eps) + self.eps)

        # the path should keep its initial length
        if self._length0 is None:
            self._length0 = length_calc(self.poses0)
        self.loss['length'] = self.traj_length_weight * torch.abs(length_calc(self.poses) - self._length0)

        return self.loss['vis'] + self.loss['l2'] + self.loss['length'] + self.loss['smooth']
