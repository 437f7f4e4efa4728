"""Device-resident optimisation loops.

`optimize_trajectory` is the reference's TrajOpt.run (/root/reference/src/trajectory_optimization.py:100-127):
Adam over (poses, quats) with two learning rates, the criterion of ModelTraj, and the early stop on visibility
and smoothness gains — but every step is a fixed sequence of kernel launches through the C ABI (visibility
forward/backward, regularisers + their analytic gradient, Adam, the early-stop rule), with no host
synchronisation until the run ends: the stop flag lives on the device and turns the remaining updates into no-ops.

The classes in model.py + torch.optim.Adam remain the drop-in path; this module is the launch-only fast path.
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr


LAST_OUTPUTS = 2   # TOHIP_TRAJ_OPT_LAST_OUTPUTS


class TrajOptResult:
    def __init__(self, steps_taken, stopped, losses, vis_gain, smooth_gain):
        self.steps_taken, self.stopped, self.losses = steps_taken, stopped, losses
        self.visibility_gain, self.smoothness_gain = vis_gain, smooth_gain


class _OptRun:
    """One device-resident optimisation run as the library sees it (struct tohip_traj_opt): B equal-length trajectories over one
    packed cloud, every per-step vector allocated once; step(i) is ONE library call — five launches (tohip_traj_opt_step)."""

    def __init__(self, models, n_opt_steps, lr_pose, lr_quat, rewards_th, smoothness_th, vis_wps_dist, betas, adam_eps):
        L = _lib.lib()
        m0 = models[0]
        self.models, self.B, self.dev, self.n_steps = models, len(models), m0.device, max(int(n_opt_steps), 1)
        B, dev = self.B, self.dev
        cloud, rig = m0._cloud, m0._rig
        W = self.W = m0.poses.shape[0]
        step_w = m0._wps_step(vis_wps_dist)
        n_eval = self.n_eval = (W + step_w - 1) // step_w
        C = rig.n_cams if rig else 1
        f32 = dict(dtype=torch.float32, device=dev)
        if B == 1:   # the Parameters themselves are updated in place
            self.poses, self.quats, self.poses0 = m0.poses.data, m0.quats.data, m0.poses0.contiguous()
        else:
            self.poses = torch.cat([m.poses.data for m in models]).contiguous()
            self.quats = torch.cat([m.quats.data for m in models]).contiguous()
            self.poses0 = torch.cat([m.poses0 for m in models]).contiguous()
        if not (self.poses.is_contiguous() and self.quats.is_contiguous() and self.poses.dtype == torch.float32 and self.quats.dtype == torch.float32):
            raise RuntimeError("optimize_trajectory: poses / quats must be contiguous float32 tensors")
        self.toff = (torch.arange(B + 1, dtype=torch.int32) * n_eval).to(dev) if B > 1 else None
        self.ws = m0._workspace(n_eval) if B == 1 else ops.TrajWorkspace(cloud, B * n_eval * C, B)
        self.pg, self.qg = torch.zeros((B * W, 3), **f32), torch.zeros((B * W, 4), **f32)
        self.lo_sum = torch.empty((B, cloud.npad), **f32)
        self.minmax = torch.empty((B * n_eval * C, 2), **f32)
        self.rewards, self.scalars = torch.empty((B, cloud.n), **f32), torch.zeros((B, 4), **f32)
        self.loss_log = torch.zeros((B, self.n_steps, 8), **f32)
        self.state_log = torch.zeros((B, self.n_steps + 1, 8), **f32)
        self.moments = [torch.zeros((B * W, 3), **f32), torch.zeros((B * W, 3), **f32), torch.zeros((B * W, 4), **f32), torch.zeros((B * W, 4), **f32)]
        nbytes = L.tohip_traj_opt_scratch_bytes(W, B)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        c = _lib.TrajOpt()
        # TOHIP_TRAJ_OPT_LAST_OUTPUTS: the rewards (and the log-odds vector) are materialised by the run's last step only — what the
        # reference publishes after its loop (trajectory_optimization.py:147-157); every step computes them all the same
        c.packed, c.n_points, c.n_wps, c.wps_step, c.flags, c.n_traj, c.n_steps = cloud.blob.data_ptr(), cloud.n, W, step_w, int(m0._flags) | LAST_OUTPUTS, B, self.n_steps
        c.traj_offsets = self.toff.data_ptr() if self.toff is not None else None
        c.cam = m0._cam.c
        if rig is not None:
            c.rig = rig.c
        c.poses, c.quats, c.poses0 = self.poses.data_ptr(), self.quats.data_ptr(), self.poses0.data_ptr()
        c.smoothness_weight, c.traj_length_weight = float(m0.smoothness_weight), float(m0.traj_length_weight)
        c.lr_pose, c.lr_quat, c.beta1, c.beta2, c.adam_eps = float(lr_pose), float(lr_quat), float(betas[0]), float(betas[1]), float(adam_eps)
        c.rewards_th, c.smoothness_th = float(rewards_th), float(smoothness_th)
        c.exp_avg_p, c.exp_avg_sq_p, c.exp_avg_q, c.exp_avg_sq_q = (t.data_ptr() for t in self.moments)
        c.poses_grad, c.quats_grad = self.pg.data_ptr(), self.qg.data_ptr()
        c.poses_grad_eval = c.quats_grad_eval = None
        c.lo_sum, c.minmax, c.rewards, c.scalars = self.lo_sum.data_ptr(), self.minmax.data_ptr(), self.rewards.data_ptr(), self.scalars.data_ptr()
        c.loss_log, c.state_log = self.loss_log.data_ptr(), self.state_log.data_ptr()
        c.workspace, c.workspace_bytes = self.ws.buf.data_ptr(), self.ws.bytes
        c.scratch, c.scratch_bytes = self.scratch.data_ptr(), nbytes
        self.c, self.ref, self.fn = c, ctypes.byref(c), L.tohip_traj_opt_step

    def run(self, n):
        idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        if n != self.n_steps:   # a run cut short has no "last step" to leave the rewards: every step writes them
            self.c.flags &= ~LAST_OUTPUTS
        with torch.cuda.device(idx):
            stream = torch._C._cuda_getCurrentRawStream(idx)
            for i in range(n):
                rc = self.fn(self.ref, i, stream)
                if rc:
                    check(rc, "tohip_traj_opt_step")
                self.ws.generation += 1
        for m in self.models:   # the Parameters (or their copies) were written through raw pointers
            torch.autograd.graph.increment_version(m.poses)
            torch.autograd.graph.increment_version(m.quats)

    def results(self, n):
        """The run's only host synchronisation: the final state rows and the loss logs."""
        st = self.state_log[:, n].cpu()
        lt = self.loss_log.cpu()
        out = []
        for b, m in enumerate(self.models):
            if self.B > 1:
                m.poses.data.copy_(self.poses[b * self.W:(b + 1) * self.W])
                m.quats.data.copy_(self.quats[b * self.W:(b + 1) * self.W])
            steps = int(st[b, 3].item())
            if steps > 0:   # (a trajectory that was never stepped keeps what it had)
                row = lt[b, steps - 1]
                m.rewards = self.rewards[b]
                m.loss = {"vis": row[0], "l2": row[1], "length": row[2], "smooth": row[3]}
            out.append(TrajOptResult(steps, bool(st[b, 2].item() != 0), lt[b, :steps, 4].tolist(), float(st[b, 4]), float(st[b, 5])))
        return out


@torch.no_grad()
def optimize_trajectory(model, n_opt_steps=10, lr_pose=0.1, lr_quat=0.0, rewards_th=1.2, smoothness_th=0.9,
                        vis_wps_dist=0.5, betas=(0.9, 0.999), adam_eps=1e-8):
    """Runs up to n_opt_steps on `model` (a ModelTraj) in place; returns a TrajOptResult (one host sync, at the end).
    model.poses / model.quats hold the optimised trajectory, model.rewards the last rewards, model.loss the last terms.

    A step is ONE library call and FIVE launches (tohip_traj_opt_step): the waypoint selection is a stride of the first launch's
    reads, the regularisers and Adam's constants are one block more of the THIRD launch (the sparse kernel's: nothing launched
    before it may read them), the parameter update and the early-stop bookkeeping are the tail of the last launch's blocks.  A waypoint-sharded or occlusion-aware model has a collective or a hull
    pass inside the step and goes through the separate calls (forward | all-reduce | reward + backward | tohip_traj_step_tail).
    (A HIP-graph replay of the step was measured slower than issuing its launches — a replay costs 10-16 us of host time by
    itself, five launches 17 us, and the GPU side is the same — so there is no graph variant.)"""
    if n_opt_steps <= 0:   # nothing to run: the model keeps its rewards and loss terms
        return TrajOptResult(0, False, [], 0.0, 0.0)
    if getattr(model, "_n_global", None) is not None:
        return _optimize_trajectory_points(model, n_opt_steps, lr_pose, lr_quat, rewards_th, smoothness_th, vis_wps_dist, betas, adam_eps)
    if model._shard.world_size > 1 or getattr(model._shard, "_always", False) or model._occlusion is not None:
        return _optimize_trajectory_split(model, n_opt_steps, lr_pose, lr_quat, rewards_th, smoothness_th, vis_wps_dist, betas, adam_eps)
    run = _OptRun([model], n_opt_steps, lr_pose, lr_quat, rewards_th, smoothness_th, vis_wps_dist, betas, adam_eps)
    run.run(n_opt_steps)
    return run.results(n_opt_steps)[0]


@torch.no_grad()
def _optimize_trajectory_points(model, n_opt_steps, lr_pose, lr_quat, rewards_th, smoothness_th, vis_wps_dist, betas, adam_eps):
    """optimize_trajectory of a POINT-sharded model (distributed.PointShard): per step the point-sharded visibility step
    (ops.PointShardStep: this rank's points, every waypoint, two small collectives) and the replicated O(W) remainder
    (tohip_traj_step_tail) — every rank holds the same gradients, so every rank takes the same step."""
    L = _lib.lib()
    dev = model.device
    W = model.poses.shape[0]
    step_w = model._wps_step(vis_wps_dist)
    n_eval = (W + step_w - 1) // step_w
    st = model._point_step(n_eval)
    f32 = dict(dtype=torch.float32, device=dev)
    pg, qg = torch.zeros((W, 3), **f32), torch.zeros((W, 4), **f32)
    loss_terms = torch.zeros((n_opt_steps + 1, 8), **f32)
    state = torch.zeros(8, **f32)
    mp, vp = torch.zeros((W, 3), **f32), torch.zeros((W, 3), **f32)
    mq, vq = torch.zeros((W, 4), **f32), torch.zeros((W, 4), **f32)
    poses, quats = model.poses.data, model.quats.data
    stride = ((step_w - 1) & 0xffff) << 8
    with torch.cuda.device(dev):
        for _ in range(n_opt_steps):
            rewards, scalars, pg_e, qg_e = st.step(poses, quats, flags_extra=stride)
            check(L.tohip_traj_step_tail(ptr(poses), ptr(quats), ptr(model.poses0), W, ptr(pg_e), ptr(qg_e), n_eval, step_w,
                                         ptr(pg), ptr(qg), ptr(mp), ptr(vp), ptr(mq), ptr(vq), float(model.smoothness_weight),
                                         float(model.traj_length_weight), float(model.eps), float(lr_pose), float(lr_quat),
                                         betas[0], betas[1], adam_eps, float(rewards_th), float(smoothness_th), ptr(scalars),
                                         ptr(loss_terms), ptr(state), stream_ptr()), "step tail")
    torch.autograd.graph.increment_version(model.poses)
    torch.autograd.graph.increment_version(model.quats)
    stt = state.cpu()  # the run's only host synchronisation
    steps = int(stt[3].item())
    lt_host = loss_terms[:max(steps, 1)].cpu()
    model.rewards = st.rewards
    model._mean_reward = st.scalars[0].clone()
    model.loss = {"vis": lt_host[-1, 0], "l2": lt_host[-1, 1], "length": lt_host[-1, 2], "smooth": lt_host[-1, 3]}
    return TrajOptResult(steps, bool(stt[2].item() != 0), lt_host[:, 4].tolist(), float(stt[4]), float(stt[5]))


@torch.no_grad()
def _optimize_trajectory_split(model, n_opt_steps, lr_pose, lr_quat, rewards_th, smoothness_th, vis_wps_dist, betas, adam_eps):
    """optimize_trajectory with a collective (waypoint sharding) or a hull pass (occlusion rows) inside the step."""
    L = _lib.lib()
    dev = model.device
    cloud, cam, rig = model._cloud, model._cam, model._rig
    W = model.poses.shape[0]
    step_w = model._wps_step(vis_wps_dist)
    n_eval = (W + step_w - 1) // step_w
    # waypoint sharding (one process per GPU): this rank evaluates rows [lo_e, hi_e) of the evaluated waypoints; the
    # log-odds vector and the (n_eval, 7) gradient rows are all-reduced, everything else is replicated
    lo_e, hi_e = model._shard.bounds(n_eval)
    n_loc = hi_e - lo_e
    ws = model._workspace(max(n_loc, 1))
    f32 = dict(dtype=torch.float32, device=dev)
    g_e = torch.zeros((n_eval, 7), **f32)  # rows outside this rank's range stay zero
    pg_e, qg_e = torch.empty((n_eval, 3), **f32), torch.empty((n_eval, 4), **f32)
    pg_loc, qg_loc = torch.empty((max(n_loc, 1), 3), **f32), torch.empty((max(n_loc, 1), 4), **f32)
    pg, qg = torch.zeros((W, 3), **f32), torch.zeros((W, 4), **f32)
    lo_sum = torch.empty(cloud.npad, **f32)
    minmax = torch.empty((max(n_loc, 1) * (rig.n_cams if rig else 1), 2), **f32)
    rewards, scalars = torch.empty(cloud.n, **f32), torch.zeros(4, **f32)
    loss_terms = torch.zeros((n_opt_steps + 1, 8), **f32)
    state = torch.zeros(8, **f32)
    gout = torch.ones(1, **f32)
    mp, vp = torch.zeros((W, 3), **f32), torch.zeros((W, 3), **f32)
    mq, vq = torch.zeros((W, 4), **f32), torch.zeros((W, 4), **f32)
    poses, quats = model.poses.data, model.quats.data
    rig_ref = rig.ref() if rig is not None else ops._NULL_RIG
    occluded = model._occlusion is not None
    # this rank's evaluated waypoints are rows lo_e * step_w, (lo_e + 1) * step_w, ... of the Parameters: read in place
    # (TOHIP_TRAJ_STRIDE in the flags: no gather launch)
    flags_fwd = int(model._flags) | (((step_w - 1) & 0xffff) << 8)
    p_at, q_at = poses[lo_e * step_w:], quats[lo_e * step_w:]

    def iteration():
        s = stream_ptr()
        occ = None
        if occluded and n_loc > 0:
            occ = model._occlusion_rows(poses[lo_e * step_w:(hi_e - 1) * step_w + 1:step_w].contiguous(),
                                        quats[lo_e * step_w:(hi_e - 1) * step_w + 1:step_w].contiguous())
        if n_loc > 0:
            check(L.tohip_traj_forward(ptr(cloud.blob), cloud.n, ptr(p_at), ptr(q_at), n_loc, cam.ref(), rig_ref,
                                       flags_fwd, ptr(occ), ptr(lo_sum), ptr(minmax), ptr(rewards), ptr(ws.buf), ws.bytes, s),
                  "forward")
            ws.generation += 1
        else:
            lo_sum.zero_()
        ops.allreduce_log_odds(model._shard, cloud, ws, lo_sum, local=n_loc > 0)
        if n_loc > 0:
            # rewards, their mean and the loss scalars share the backward's first launch
            check(L.tohip_traj_reward_backward(ptr(cloud.blob), cloud.n, n_loc, cam.ref(), rig_ref, model._flags, ptr(occ), ptr(lo_sum),
                                               cam.eps, 1, ptr(rewards), ptr(scalars), ptr(gout), ptr(pg_loc), ptr(qg_loc), ptr(ws.buf),
                                               ws.bytes, s), "reward + backward")
        else:
            check(L.tohip_traj_reward(ptr(cloud.blob), ptr(lo_sum), cloud.n, cam.eps, 0, ptr(rewards), ptr(scalars), ptr(ws.buf), ws.bytes, s),
                  "reward")
        # assemble every rank's gradient rows: ONE (n_eval, 7) all-reduce, then the replicated remainder of the step
        if n_loc > 0:
            g_e[lo_e:hi_e, :3], g_e[lo_e:hi_e, 3:] = pg_loc, qg_loc
        model._shard.allreduce_sum(g_e)
        pg_e.copy_(g_e[:, :3])
        qg_e.copy_(g_e[:, 3:])
        g_e.zero_()  # the other ranks' rows must be zero again before the next sum
        # the O(W) remainder of the step in one launch: scatter, regularisers, both Adam updates, early stop
        check(L.tohip_traj_step_tail(ptr(poses), ptr(quats), ptr(model.poses0), W, ptr(pg_e), ptr(qg_e), n_eval, step_w,
                                     ptr(pg), ptr(qg), ptr(mp), ptr(vp), ptr(mq), ptr(vq), float(model.smoothness_weight),
                                     float(model.traj_length_weight), float(model.eps), float(lr_pose), float(lr_quat),
                                     betas[0], betas[1], adam_eps, float(rewards_th), float(smoothness_th), ptr(scalars),
                                     ptr(loss_terms), ptr(state), s), "step tail")

    with torch.cuda.device(dev):
        for _ in range(n_opt_steps):
            iteration()
    st = state.cpu()  # the run's only host synchronisation
    steps = int(st[3].item())
    lt_host = loss_terms[:max(steps, 1)].cpu()
    model.rewards = rewards
    model.loss = {"vis": lt_host[-1, 0], "l2": lt_host[-1, 1], "length": lt_host[-1, 2], "smooth": lt_host[-1, 3]}
    return TrajOptResult(steps, bool(st[2].item() != 0), lt_host[:, 4].tolist(), float(st[4]), float(st[5]))


@torch.no_grad()
def optimize_trajectories(models, n_opt_steps=10, lr_pose=0.1, lr_quat=0.0, rewards_th=1.2, smoothness_th=0.9,
                          vis_wps_dist=0.5, betas=(0.9, 0.999), adam_eps=1e-8):
    """`optimize_trajectory` for several candidate trajectories over the SAME cloud at once: every step is one
    set of launches for all of them — their evaluated waypoints go through the visibility kernels as one batch of virtual
    waypoints, each trajectory keeps its own log-odds vector, rewards, loss terms, Adam moments and early-stop state (a
    trajectory that has stopped stays put while the others go on).  Each model ends up exactly — bit for bit — where its own
    `optimize_trajectory` run would have put it.  Models: ModelTraj built on the same points with the same camera, rig and
    mode, equal numbers of waypoints and the same waypoint step; no sharding, no occlusion.  -> [TrajOptResult]."""
    m0 = models[0]
    cloud, cam, rig = m0._cloud, m0._cam, m0._rig
    W = m0.poses.shape[0]
    step_w = m0._wps_step(vis_wps_dist)
    for m in models:
        if (m.poses.shape[0] != W or m._wps_step(vis_wps_dist) != step_w or m._cloud.n != cloud.n or m._flags != m0._flags or
                (m._rig is None) != (rig is None) or m._shard.world_size > 1 or m._occlusion is not None or
                bytes(m._cam.c) != bytes(cam.c) or m.smoothness_weight != m0.smoothness_weight or
                m.traj_length_weight != m0.traj_length_weight):
            raise ValueError("optimize_trajectories: the models must share the cloud, camera, rig, mode, waypoint count and step")
        if m is not m0 and m.points.data_ptr() != m0.points.data_ptr() and not torch.equal(m.points, m0.points):
            raise ValueError("optimize_trajectories: the models must be built on the same points")
        if m is not m0 and (m.device != m0.device or float(m.eps) != float(m0.eps)):
            raise ValueError("optimize_trajectories: the models must live on one device and share eps")
        if m is not m0 and rig is not None and (m._rig.n_cams != rig.n_cams or not torch.equal(m._rig.q, rig.q) or not torch.equal(m._rig.t, rig.t)):
            raise ValueError("optimize_trajectories: the models must share the camera rig (extrinsics differ)")
    if n_opt_steps <= 0:   # nothing to run: the models keep their rewards and loss terms
        return [TrajOptResult(0, False, [], 0.0, 0.0) for _ in models]
    run = _OptRun(list(models), n_opt_steps, lr_pose, lr_quat, rewards_th, smoothness_th, vis_wps_dist, betas, adam_eps)
    run.run(n_opt_steps)
    return run.results(n_opt_steps)


class PoseOptResult:
    def __init__(self, losses):
        self.losses = losses


@torch.no_grad()
def optimize_pose(model, n_opt_steps=100, lr_pose=0.1, lr_quat=0.1, hpr=False, betas=(0.9, 0.999), adam_eps=1e-8):
    """The reference's PoseOpt loop (/root/reference/src/pose_optimization.py:93-97,124-141): n_opt_steps of
    `loss = model(hpr); loss.backward(); Adam(trans @ lr_pose, quat @ lr_quat).step()` on a ModelPose, in place, as
    launches only — two per step (tohip_pose_opt_step: ONE pass over the cloud for observations, loss and gradient sums, then a
    one-block finish with both Adam updates and the loss log) — with one host synchronisation when the run ends.  model.trans / model.quat hold the optimised pose (the
    reference normalises the quaternion only when publishing, :102), model.observations the last observations."""
    L = _lib.lib()
    dev = model.device
    cloud, cam, ws = model._cloud, model._cam, model._ws
    f32 = dict(dtype=torch.float32, device=dev)
    mask = None
    if hpr:
        model(hpr=True)  # builds (and caches) the world-frame occlusion mask of model.py:114
        mask = model._occlusion_mask
    obs, scalars = torch.empty(cloud.n, **f32), torch.zeros(4, **f32)
    tg, qg = torch.empty((1, 3), **f32), torch.empty((1, 4), **f32)
    mt, vt = torch.zeros(3, **f32), torch.zeros(3, **f32)
    mq, vq = torch.zeros(4, **f32), torch.zeros(4, **f32)
    losses = torch.empty(max(n_opt_steps, 1), **f32)
    trans, quat = model.trans.data, model.quat.data
    fn = L.tohip_pose_opt_step
    args = (cloud.blob.data_ptr(), cloud.n, trans.data_ptr(), quat.data_ptr(), cam.ref(), mask.data_ptr() if mask is not None else None,
            obs.data_ptr(), scalars.data_ptr(), tg.data_ptr(), qg.data_ptr(), mt.data_ptr(), vt.data_ptr(), mq.data_ptr(), vq.data_ptr(),
            float(lr_pose), float(lr_quat), float(betas[0]), float(betas[1]), float(adam_eps))
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        stream = torch._C._cuda_getCurrentRawStream(idx)
        for i in range(n_opt_steps):
            # one pass over the cloud (observations, their sum, the gradient sums) and its one-block finish (loss, gradient, both
            # Adam updates, the loss log): two launches per step
            rc = fn(*args, i + 1, losses.data_ptr(), ws.buf.data_ptr(), ws.bytes, stream)
            if rc:
                check(rc, "tohip_pose_opt_step")
    torch.autograd.graph.increment_version(model.trans)
    torch.autograd.graph.increment_version(model.quat)
    model.observations = obs
    return PoseOptResult(losses[:n_opt_steps].cpu().tolist())  # the run's only host synchronisation


def _adam_update(L, entries, arr):
    """One launch for all the listed (group, param, state, grad) entries (at most ADAM_MAX_GROUPS per launch)."""
    k = 0
    dev = None
    for group, p, st, g in entries:
        e = arr[k]
        e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        e.n = p.numel()
        e.lr, (e.beta1, e.beta2), e.eps, e.step = group["lr"], group["betas"], group["eps"], int(st["step"])
        k += 1
        if dev is not None and p.device != dev:
            raise RuntimeError("one optimizer step over parameters of several devices is not supported")
        dev = p.device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if torch.cuda.current_device() == idx:
        rc = L.tohip_adam_step_multi(arr, k, torch._C._cuda_getCurrentRawStream(idx))
    else:
        with torch.cuda.device(idx):
            rc = L.tohip_adam_step_multi(arr, k, torch._C._cuda_getCurrentRawStream(idx))
    if rc:
        check(rc, "tohip_adam_step_multi")
    for _, q, _, _ in entries:   # the kernel wrote through raw pointers: autograd's in-place guards must still see an update
        torch.autograd.graph.increment_version(q)


def _steppable(p, g):
    return p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32 and g.is_contiguous() and not g.is_sparse


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam for the models' Parameters (defaults of the reference's loops: betas (0.9, 0.999), eps 1e-8, no
    weight decay, no amsgrad) with ONE kernel launch for all parameters (tohip_adam_step_multi) instead of the dozen small foreach
    kernels per group — the drop-in loop is launch-bound.  Same constructor (parameter groups with their own `lr`), so
    `ExponentialLR` and friends work unchanged:

        optimizer = Adam([{'params': [model.poses], 'lr': 0.1}, {'params': [model.quats], 'lr': 0.02}])
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._arr = (_lib.AdamGroup * _lib.ADAM_MAX_GROUPS)()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries = []
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not g.is_contiguous():
                    g = g.contiguous()
                if not _steppable(p, g):
                    raise RuntimeError("trajectory_optimization_amd.optimizer.Adam steps contiguous float32 HIP tensors only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                entries.append((group, p, st, g))
        L = _lib.lib()
        for i in range(0, len(entries), _lib.ADAM_MAX_GROUPS):
            _adam_update(L, entries[i:i + _lib.ADAM_MAX_GROUPS], self._arr)
        return loss


# ---- torch.optim.Adam itself, for the models' Parameters ----------------------------------------------------------------------
# The reference builds `torch.optim.Adam([{'params': [model.poses], 'lr': lr_pose}, {'params': [model.quats], 'lr': lr_quat}])`
# (/root/reference/src/trajectory_optimization.py:91-94).  Its step() is ~14 foreach launches and 0.13-0.15 ms of host time for
# these two small tensors — longer than everything else in the loop together.  OPT-IN (r06; until r05 the hooks were installed
# process-wide when the first model was built — a drop-in must not change global torch state on its own): after
#     accelerate_torch_adam(opt)            # this optimizer instance only (Optimizer.register_step_pre_hook / _post_hook), or
#     accelerate_torch_adam(True)           # every torch.optim.Adam of the process (the global hook registry), or
#     ModelTraj(..., fast_adam=True)        # = accelerate_torch_adam(True), said where the model is built
# a step pre-hook updates the Parameters that belong to a model of this package (tagged at construction) with the one-launch
# kernel, on torch's own state entries (`step`, `exp_avg`, `exp_avg_sq`: state_dict(), schedulers and a later switch back all keep
# working), and hides their gradients from torch's step for its duration; every other parameter, and every optimizer configuration
# other than plain Adam (amsgrad, weight decay, maximize, capturable, fused, differentiable, a closure) is left to torch.
# Nothing is registered anywhere until one of the three is called.

_ACCEL = {"on": False, "installed": False}


def accelerate_torch_adam(enable=True):
    """enable = a torch.optim.Adam INSTANCE: take over the update of tagged Parameters inside that optimizer's step() (hooks on the
    instance; returns it).  enable = True / False: the same for every torch.optim.Adam of the process, on / off (the process-wide
    hooks are registered on the first True and do nothing while off).  Off and nowhere registered by default."""
    if isinstance(enable, torch.optim.Optimizer):
        opt = enable
        if not opt.__dict__.get("_tohip_accel"):
            opt.register_step_pre_hook(_adam_pre_hook_instance)
            opt.register_step_post_hook(_adam_post_hook)
            opt.__dict__["_tohip_accel"] = True
        return opt
    _ACCEL["on"] = bool(enable)
    if enable:
        _install_hooks()
    return None


def torch_adam_accelerated():
    """-> (process-wide switch on?, process-wide hooks registered?)"""
    return _ACCEL["on"], _ACCEL["installed"]


def tag_parameter(p):
    """Mark a Parameter as one whose plain-Adam update MAY be taken over once the user asks for it (the models call this for
    poses / quats / trans / quat).  Registers nothing."""
    p._tohip_param = True
    return p


def _install_hooks():
    if not _ACCEL["installed"]:
        from torch.optim.optimizer import register_optimizer_step_pre_hook, register_optimizer_step_post_hook
        register_optimizer_step_pre_hook(_adam_pre_hook)
        register_optimizer_step_post_hook(_adam_post_hook)
        _ACCEL["installed"] = True


def _plain_adam_group(g):
    return not (g.get("amsgrad", False) or g.get("weight_decay", 0) != 0 or g.get("maximize", False) or g.get("capturable", False) or
                g.get("differentiable", False) or g.get("fused", False) or g.get("decoupled_weight_decay", False) or
                not isinstance(g["lr"], float))


def _restore_stash(opt):
    stash = opt.__dict__.pop("_tohip_adam_stash", None)
    if stash:
        for _, p, _, g in stash:
            if p.grad is None:
                p.grad = g


def _adam_pre_hook_instance(opt, args, kwargs):
    return _adam_pre_hook(opt, args, kwargs, force=True)


def _adam_pre_hook(opt, args, kwargs, force=False):
    _restore_stash(opt)   # a step() that raised between the two hooks left the gradients hidden: put them back first
    if (not force and (not _ACCEL["on"] or opt.__dict__.get("_tohip_accel"))) or type(opt) is not torch.optim.Adam or (len(args) > 1 and args[1] is not None) or kwargs.get("closure") is not None:   # args[0] is the optimizer
        return None
    entries = []
    for group in opt.param_groups:
        plain = None
        for p in group["params"]:
            g = p.grad
            if g is None or not getattr(p, "_tohip_param", False):
                continue
            if plain is None:
                plain = _plain_adam_group(group)
            if not plain or not _steppable(p, g):
                continue
            st = opt.state[p]
            if len(st) == 0:   # what torch.optim.Adam._init_group creates (step on the host: neither capturable nor fused)
                st["step"] = torch.tensor(0.0, dtype=torch.float64 if torch.get_default_dtype() == torch.float64 else torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif st["step"].is_cuda or not st["exp_avg"].is_contiguous():
                continue
            entries.append((group, p, st, g))
    if not entries:
        return None
    arr = opt.__dict__.get("_tohip_adam_arr")
    if arr is None:
        arr = opt.__dict__["_tohip_adam_arr"] = (_lib.AdamGroup * _lib.ADAM_MAX_GROUPS)()
    L = _lib.lib()
    with torch.no_grad():
        for _, _, st, _ in entries:
            st["step"] += 1
        for i in range(0, len(entries), _lib.ADAM_MAX_GROUPS):
            _adam_update(L, entries[i:i + _lib.ADAM_MAX_GROUPS], arr)
    for _, p, _, _ in entries:
        p.grad = None    # torch's step skips parameters without a gradient; the post-hook puts it back
    opt.__dict__["_tohip_adam_stash"] = entries
    return None


def _adam_post_hook(opt, args, kwargs):
    _restore_stash(opt)
    return None
