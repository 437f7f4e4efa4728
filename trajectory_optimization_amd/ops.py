"""Thin torch-tensor wrappers over the C ABI (include/trajopt_hip.h).

PyTorch here is plumbing only: device memory, streams and autograd bookkeeping.  Every numeric
step of the hot path runs in libtrajopt_hip.so; nothing in this module computes on the CPU and
nothing falls back to torch ops when the library is missing.
"""
import ctypes

import numpy as np

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _dev_f32(t, device):
    return torch.as_tensor(t, dtype=torch.float32, device=device).contiguous()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on a HIP device (got {t.device}); the visibility path has no CPU fallback")


class PackedCloud:
    """The cloud in the kernels' layout (x|y|z, padded); built once per CLOUD (it is constant over an optimisation run:
    /root/reference/src/model.py:80,174) and shared by every model over it — the reference builds a model per message pair over the
    same map (/root/reference/src/trajectory_optimization.py:129-136): `ModelTraj(cloud, ...)` / `ModelTraj(points, ..., cloud=other)`
    take a packed cloud as it is.  `points` keeps the caller's (N,3) f32 tensor (the models' `.points`)."""

    def __init__(self, points, sort=True):
        _require_cuda(points, "points")
        pts = points.detach().to(torch.float32).contiguous()
        if pts.dim() != 2 or pts.shape[1] != 3 or pts.shape[0] == 0:
            raise ValueError(f"points must be (N,3) with N>0, got {tuple(pts.shape)}")
        L = _lib.lib()
        self.points, self.sorted = pts, bool(sort)
        self.n = pts.shape[0]
        self.npad = L.tohip_padded_points(self.n)
        self.device = pts.device
        self.blob = torch.empty(L.tohip_packed_cloud_bytes(self.n), dtype=torch.uint8, device=pts.device)
        wsb = L.tohip_pack_workspace_bytes(self.n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=pts.device)
        with torch.cuda.device(pts.device):
            check(L.tohip_pack_cloud(ptr(pts), self.n, int(bool(sort)), ptr(self.blob), ptr(ws), wsb, stream_ptr()),
                  "tohip_pack_cloud")
        # views into the blob (for tests / debugging): sorted x|y|z and the permutation to the caller's order
        self.soa = self.blob[:12 * self.npad].view(torch.float32)
        self.perm = self.blob[12 * self.npad:16 * self.npad].view(torch.int32)
        inv0 = 16 * self.npad + 16 * (self.npad // 256)
        self.inv_perm = self.blob[inv0:inv0 + 4 * self.n].view(torch.int32)  # caller's index -> sorted position


def _sorted_rows(cloud):
    """The cloud's points as (N,3) rows in the PACKED order (row s = the caller's row perm[s]); built on first use, kept with the cloud
    (12 bytes per point).  The occlusion refresh culls THESE rows: a waypoint's kept indices are then positions of the packed order,
    ascending — its bit row is written run by run instead of bit by scattered bit, and the hull pass's gather walks memory in order."""
    t = getattr(cloud, "_sorted_rows", None)
    if t is None:
        t = cloud.points[cloud.perm[:cloud.n].long()].contiguous() if cloud.sorted else cloud.points
        cloud._sorted_rows = t
    return t


class Camera:
    """Host-side camera constants (struct tohip_camera)."""

    def __init__(self, K, img_width, img_height, min_dist=1.0, max_dist=5.0, eps=1e-6):
        Kh = torch.as_tensor(K, dtype=torch.float32).detach().cpu().reshape(9).tolist()
        self.c = _lib.make_camera(Kh, img_width, img_height, min_dist, max_dist, eps)
        self.eps = float(eps)

    def ref(self):
        return ctypes.byref(self.c)


class CameraRig:
    """Multi-camera rig extrinsics on the device (struct tohip_rig)."""

    def __init__(self, rig_quats, rig_trans, device):
        self.q = _dev_f32(rig_quats, device)
        self.t = _dev_f32(rig_trans, device) if rig_trans is not None else torch.zeros_like(self.q[:, :3]).contiguous()
        self.n_cams = self.q.shape[0]
        self.c = _lib.Rig(self.n_cams, self.q.data_ptr(), self.t.data_ptr())

    def ref(self):
        return ctypes.byref(self.c)


_NULL_RIG = ctypes.POINTER(_lib.Rig)()


class TrajWorkspace:
    """Scratch + step state of the ModelTraj kernels.  Zero-filled once (the C ABI's contract); `generation` counts the
    forwards that used it — the backward of a step must find the state its own forward left (model.py checks)."""

    def __init__(self, cloud, n_virtual, n_traj=1):
        self.bytes = _lib.lib().tohip_traj_workspace_bytes_multi(cloud.n, n_virtual, n_traj)
        self.buf = torch.zeros(self.bytes, dtype=torch.uint8, device=cloud.device)
        self.n_virtual, self.n_traj = n_virtual, n_traj
        self.generation = 0


DENSE = 1  # TOHIP_TRAJ_DENSE


def traj_forward(cloud, poses, quats, cam, ws, rig=None, flags=0, occ=None, lo_sum=None, minmax=None, rewards_half=None):
    """-> (lo_sum[npad] in packed order (first N valid), minmax[V,2]) for the given waypoints (this rank's shard).
    Leaves the step's state in `ws` for traj_backward.  rewards_half: optional (N,) f32 tensor filled with 0.5 on the way
    (hand it to traj_reward as `rewards=` with prefilled=True)."""
    W = poses.shape[0]
    C = rig.n_cams if rig is not None else 1
    if lo_sum is None:
        lo_sum = torch.empty(cloud.npad, dtype=torch.float32, device=cloud.device)
    if minmax is None:
        minmax = torch.empty((W * C, 2), dtype=torch.float32, device=cloud.device)
    ws.generation += 1
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_forward(ptr(cloud.blob), cloud.n, ptr(poses), ptr(quats), W, cam.ref(),
                                            rig.ref() if rig is not None else _NULL_RIG, int(flags), ptr(occ), ptr(lo_sum), ptr(minmax),
                                            ptr(rewards_half), ptr(ws.buf), ws.bytes, stream_ptr()), "tohip_traj_forward")
    return lo_sum, minmax


def traj_reward(cloud, lo_sum, cam, ws, rewards=None, scalars=None, prefilled=False):
    """-> (rewards[N], scalars[4] = mean, loss_vis, dloss/dreward, -).  prefilled: `rewards` holds 0.5 everywhere (traj_forward's
    rewards_half): only the others are stored."""
    if rewards is None:
        rewards = torch.empty(cloud.n, dtype=torch.float32, device=cloud.device)
    if scalars is None:
        scalars = torch.empty(4, dtype=torch.float32, device=cloud.device)  # all four written by the kernel
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_reward(ptr(cloud.blob), ptr(lo_sum), cloud.n, cam.eps, int(bool(prefilled)), ptr(rewards), ptr(scalars),
                                           ptr(ws.buf), ws.bytes, stream_ptr()), "tohip_traj_reward")
    return rewards, scalars


def traj_backward(cloud, n_wps, cam, ws, lo_sum, grad_rewards=None, scalars=None, gout=None, rig=None, flags=0, occ=None):
    """Gradients of the step whose traj_forward last used `ws` (same cloud, n_wps, rig, flags, occ).
    lo_sum: the (all-reduced) log-odds vector in packed order, as returned by traj_forward."""
    pg = torch.empty((n_wps, 3), dtype=torch.float32, device=cloud.device)
    qg = torch.empty((n_wps, 4), dtype=torch.float32, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_backward(ptr(cloud.blob), cloud.n, n_wps, cam.ref(),
                                             rig.ref() if rig is not None else _NULL_RIG, int(flags), ptr(occ), ptr(lo_sum),
                                             ptr(grad_rewards), ptr(scalars), ptr(gout), ptr(pg), ptr(qg), ptr(ws.buf), ws.bytes,
                                             stream_ptr()), "tohip_traj_backward")
    return pg, qg


def traj_reward_backward(cloud, n_wps, cam, ws, lo_sum, gout, rewards=None, prefilled=False, rig=None, flags=0, occ=None):
    """traj_reward + traj_backward of the fused visibility loss in two launches instead of three.
    -> (rewards[N], scalars[4], poses_grad (n_wps,3), quats_grad (n_wps,4))."""
    if rewards is None:
        rewards = torch.empty(cloud.n, dtype=torch.float32, device=cloud.device)
    scalars = torch.empty(4, dtype=torch.float32, device=cloud.device)
    pg = torch.empty((n_wps, 3), dtype=torch.float32, device=cloud.device)
    qg = torch.empty((n_wps, 4), dtype=torch.float32, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_reward_backward(ptr(cloud.blob), cloud.n, n_wps, cam.ref(), rig.ref() if rig is not None else _NULL_RIG,
                                                    int(flags), ptr(occ), ptr(lo_sum), cam.eps, int(bool(prefilled)), ptr(rewards),
                                                    ptr(scalars), ptr(gout), ptr(pg), ptr(qg), ptr(ws.buf), ws.bytes, stream_ptr()),
              "tohip_traj_reward_backward")
    return rewards, scalars, pg, qg


def traj_forward_backward(cloud, poses, quats, cam, ws, gout, rig=None, flags=0, occ=None, lo_sum=None, minmax=None, rewards=None):
    """The whole step of the fused visibility loss when no collective sits between forward and backward (tohip_traj_forward_backward,
    five launches).  -> (rewards[N], scalars[4], poses_grad (W,3), quats_grad (W,4), lo_sum[npad] packed order, minmax[V,2])."""
    W = poses.shape[0]
    C = rig.n_cams if rig is not None else 1
    dev = cloud.device
    if lo_sum is None:
        lo_sum = torch.empty(cloud.npad, dtype=torch.float32, device=dev)
    if minmax is None:
        minmax = torch.empty((W * C, 2), dtype=torch.float32, device=dev)
    if rewards is None:
        rewards = torch.empty(cloud.n, dtype=torch.float32, device=dev)
    scalars = torch.empty(4, dtype=torch.float32, device=dev)
    pg = torch.empty((W, 3), dtype=torch.float32, device=dev)
    qg = torch.empty((W, 4), dtype=torch.float32, device=dev)
    ws.generation += 1
    with torch.cuda.device(dev):
        check(_lib.lib().tohip_traj_forward_backward(ptr(cloud.blob), cloud.n, ptr(poses), ptr(quats), W, cam.ref(),
                                                     rig.ref() if rig is not None else _NULL_RIG, int(flags), ptr(occ), ptr(lo_sum), ptr(minmax),
                                                     ptr(rewards), ptr(scalars), ptr(gout), ptr(pg), ptr(qg), ptr(ws.buf), ws.bytes, stream_ptr()),
              "tohip_traj_forward_backward")
    return rewards, scalars, pg, qg, lo_sum, minmax


def traj_forward_backward_multi(cloud, poses, quats, traj_offsets, cam, ws, gout, rig=None, flags=0, lo_sum=None, minmax=None, rewards=None):
    """traj_forward_backward for B trajectories laid end to end (traj_forward_multi's layout; gout: (B,) dL/d loss_vis each).
    -> (rewards (B,N), scalars (B,4), poses_grad (W,3), quats_grad (W,4), lo_sum (B,npad), minmax (V,2))."""
    W, B = poses.shape[0], traj_offsets.numel() - 1
    C = rig.n_cams if rig is not None else 1
    dev = cloud.device
    if lo_sum is None:
        lo_sum = torch.empty((B, cloud.npad), dtype=torch.float32, device=dev)
    if minmax is None:
        minmax = torch.empty((W * C, 2), dtype=torch.float32, device=dev)
    if rewards is None:
        rewards = torch.empty((B, cloud.n), dtype=torch.float32, device=dev)
    scalars = torch.empty((B, 4), dtype=torch.float32, device=dev)
    pg = torch.empty((W, 3), dtype=torch.float32, device=dev)
    qg = torch.empty((W, 4), dtype=torch.float32, device=dev)
    ws.generation += 1
    with torch.cuda.device(dev):
        check(_lib.lib().tohip_traj_forward_backward_multi(ptr(cloud.blob), cloud.n, ptr(poses), ptr(quats), W, ptr(traj_offsets), B, cam.ref(),
                                                           rig.ref() if rig is not None else _NULL_RIG, int(flags), None, ptr(lo_sum), ptr(minmax),
                                                           ptr(rewards), ptr(scalars), ptr(gout), ptr(pg), ptr(qg), ptr(ws.buf), ws.bytes,
                                                           stream_ptr()), "tohip_traj_forward_backward_multi")
    return rewards, scalars, pg, qg, lo_sum, minmax


def traj_reward_backward_multi(cloud, n_wps, n_traj, cam, ws, lo_sum, gout, rewards=None, prefilled=False, rig=None, flags=0):
    """-> (rewards (B,N), scalars (B,4), poses_grad (W,3), quats_grad (W,4)) of B trajectories (traj_forward_multi's lo_sum)."""
    if rewards is None:
        rewards = torch.empty((n_traj, cloud.n), dtype=torch.float32, device=cloud.device)
    scalars = torch.empty((n_traj, 4), dtype=torch.float32, device=cloud.device)
    pg = torch.empty((n_wps, 3), dtype=torch.float32, device=cloud.device)
    qg = torch.empty((n_wps, 4), dtype=torch.float32, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_reward_backward_multi(ptr(cloud.blob), cloud.n, n_wps, n_traj, cam.ref(),
                                                          rig.ref() if rig is not None else _NULL_RIG, int(flags), None, ptr(lo_sum),
                                                          cam.eps, int(bool(prefilled)), ptr(rewards), ptr(scalars), ptr(gout), ptr(pg),
                                                          ptr(qg), ptr(ws.buf), ws.bytes, stream_ptr()),
              "tohip_traj_reward_backward_multi")
    return rewards, scalars, pg, qg


def traj_forward_multi(cloud, poses, quats, traj_offsets, cam, ws, rig=None, flags=0, lo_sum=None, minmax=None, rewards_half=None):
    """Several trajectories over one cloud in one pass: `poses` (W,3) / `quats` (W,4) hold their waypoints end to end,
    `traj_offsets` (B+1 int32 on the device) where each starts.  -> (lo_sum (B, npad), minmax (V, 2)); ws = TrajWorkspace(cloud, V, B)."""
    W, B = poses.shape[0], traj_offsets.numel() - 1
    C = rig.n_cams if rig is not None else 1
    if lo_sum is None:
        lo_sum = torch.empty((B, cloud.npad), dtype=torch.float32, device=cloud.device)
    if minmax is None:
        minmax = torch.empty((W * C, 2), dtype=torch.float32, device=cloud.device)
    ws.generation += 1
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_forward_multi(ptr(cloud.blob), cloud.n, ptr(poses), ptr(quats), W, ptr(traj_offsets), B, cam.ref(),
                                                  rig.ref() if rig is not None else _NULL_RIG, int(flags), None, ptr(lo_sum), ptr(minmax),
                                                  ptr(rewards_half), ptr(ws.buf), ws.bytes, stream_ptr()), "tohip_traj_forward_multi")
    return lo_sum, minmax


def traj_reward_multi(cloud, lo_sum, cam, ws, rewards=None, scalars=None, prefilled=False):
    """-> (rewards (B, N), scalars (B, 4)) for the B log-odds vectors of traj_forward_multi."""
    B = lo_sum.shape[0]
    if rewards is None:
        rewards = torch.empty((B, cloud.n), dtype=torch.float32, device=cloud.device)
    if scalars is None:
        scalars = torch.empty((B, 4), dtype=torch.float32, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_reward_multi(ptr(cloud.blob), ptr(lo_sum), cloud.n, B, cam.eps, int(bool(prefilled)), ptr(rewards),
                                                 ptr(scalars), ptr(ws.buf), ws.bytes, stream_ptr()), "tohip_traj_reward_multi")
    return rewards, scalars


def traj_backward_multi(cloud, n_wps, n_traj, cam, ws, lo_sum, grad_rewards=None, scalars=None, gout=None, rig=None, flags=0):
    """Gradients (W,3), (W,4) of all trajectories' waypoints; gout: (B,) dL/d loss_vis per trajectory."""
    pg = torch.empty((n_wps, 3), dtype=torch.float32, device=cloud.device)
    qg = torch.empty((n_wps, 4), dtype=torch.float32, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_backward_multi(ptr(cloud.blob), cloud.n, n_wps, n_traj, cam.ref(),
                                                   rig.ref() if rig is not None else _NULL_RIG, int(flags), None, ptr(lo_sum),
                                                   ptr(grad_rewards), ptr(scalars), ptr(gout), ptr(pg), ptr(qg), ptr(ws.buf), ws.bytes,
                                                   stream_ptr()), "tohip_traj_backward_multi")
    return pg, qg


class PointShardStep:
    """One point-sharded visibility step (tohip_traj_pshard_*; distributed.PointShard): this rank's part of the cloud, all the
    waypoints, two small collectives.  Buffers are allocated once; step(poses, quats) -> (rewards of this rank's points, scalars
    (4: mean reward of ALL points, loss_vis, d loss_vis / d reward, -), poses_grad (W,3), quats_grad (W,4)) — scalars and
    gradients identical on every rank (gradients for dL/d loss_vis = 1)."""

    def __init__(self, cloud, n_global, n_wps, cam, ws, shard, rig=None, flags=0):
        L = _lib.lib()
        self.cloud, self.n_global, self.n_wps, self.cam, self.ws, self.shard, self.rig, self.flags = cloud, int(n_global), int(n_wps), cam, ws, shard, rig, int(flags)
        dev = cloud.device
        C = rig.n_cams if rig is not None else 1
        V = n_wps * C
        f32 = dict(dtype=torch.float32, device=dev)
        self.lo_sum, self.minmax = torch.empty(cloud.npad, **f32), torch.empty((V, 2), **f32)
        self.rewards, self.scalars = torch.empty(cloud.n, **f32), torch.empty(4, **f32)
        self.pg, self.qg = torch.empty((n_wps, 3), **f32), torch.empty((n_wps, 4), **f32)
        self.partial = torch.empty(L.tohip_traj_pshard_partial_count(V), dtype=torch.float64, device=dev)
        words, n_words = ctypes.c_void_p(), ctypes.c_int64()
        check(L.tohip_traj_extrema_view(cloud.n, V, ptr(ws.buf), ws.bytes, ctypes.byref(words), ctypes.byref(n_words)), "tohip_traj_extrema_view")
        off = words.value - ws.buf.data_ptr()
        self.extrema = ws.buf[off:off + 4 * n_words.value].view(torch.int32)   # the workspace's own words: reduced in place
        self.rig_ref = rig.ref() if rig is not None else _NULL_RIG

    def step(self, poses, quats, flags_extra=0):
        """flags_extra: TOHIP_TRAJ_STRIDE bits (the evaluated waypoints as every step-th row of poses / quats, read in place)."""
        L, c, ws = _lib.lib(), self.cloud, self.ws
        with torch.cuda.device(c.device):
            s = stream_ptr()
            check(L.tohip_traj_pshard_pass1(ptr(c.blob), c.n, self.n_global, ptr(poses), ptr(quats), self.n_wps, self.cam.ref(), self.rig_ref,
                                            self.flags | int(flags_extra), None, ptr(self.lo_sum), ptr(self.rewards), ptr(ws.buf), ws.bytes, s), "tohip_traj_pshard_pass1")
            ws.generation += 1
            self.shard.allreduce_max(self.extrema)      # collective 1: the waypoints' extrema over all points (16 B per virtual waypoint)
            check(L.tohip_traj_pshard_local(ptr(c.blob), c.n, self.n_global, self.n_wps, self.cam.ref(), self.rig_ref, self.flags, None,
                                            ptr(self.lo_sum), ptr(self.minmax), ptr(self.rewards), ptr(self.partial), ptr(ws.buf), ws.bytes, s),
                  "tohip_traj_pshard_local")
            self.shard.allreduce_sum(self.partial)      # collective 2: the sums everything after is linear in (40 doubles per virtual waypoint)
            check(L.tohip_traj_pshard_finish(c.n, self.n_global, self.n_wps, self.cam.ref(), self.rig_ref, ptr(self.partial), None,
                                             ptr(self.scalars), ptr(self.pg), ptr(self.qg), ptr(ws.buf), ws.bytes, s), "tohip_traj_pshard_finish")
        return self.rewards, self.scalars, self.pg, self.qg


def allreduce_log_odds(shard, cloud, ws, lo_sum, local=True):
    """The one data-path collective of a waypoint-sharded step: the sum of the ranks' partial log-odds vectors, in
    place.  With shard.compact only the slots some rank's forward listed as candidates travel (tohip_traj_candidate_flags ...
    tohip_slots_pack): the vector is exactly zero elsewhere on every rank.  local=False: this rank ran no forward over `ws` (it
    holds no waypoint): its vector is zero and it lists nothing."""
    if not getattr(shard, "compact", False) or shard.world_size == 1 and not shard._always:
        return shard.allreduce_sum(lo_sum)
    L = _lib.lib()
    dev = cloud.device
    nslots = cloud.npad // 256
    flags = torch.zeros(nslots, dtype=torch.int32, device=dev)
    prefix = torch.empty(nslots + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        if local:
            check(L.tohip_traj_candidate_flags(cloud.n, ws.n_virtual, ws.n_traj, ptr(ws.buf), ws.bytes, ptr(flags), stream_ptr()), "tohip_traj_candidate_flags")
        shard.allreduce_max(flags)
        check(L.tohip_slot_flags_prefix(ptr(flags), cloud.n, ptr(prefix), stream_ptr()), "tohip_slot_flags_prefix")
        count = int(prefix[nslots].item())   # the step's one host read: the union's size is the message's
        if count == 0:
            return lo_sum
        buf = getattr(ws, "_compact", None)
        if buf is None or buf.numel() < count * 256:
            buf = ws._compact = torch.empty(max(count * 256 * 2, 1 << 16), dtype=torch.float32, device=dev)
        check(L.tohip_slots_pack(ptr(flags), ptr(prefix), cloud.n, ptr(lo_sum), ptr(buf), count, 1, stream_ptr()), "tohip_slots_pack")
        shard.allreduce_sum(buf[:count * 256])
        check(L.tohip_slots_pack(ptr(flags), ptr(prefix), cloud.n, ptr(lo_sum), ptr(buf), count, 0, stream_ptr()), "tohip_slots_pack")
    return lo_sum


def traj_step_stats(cloud, ws):
    """What the last forward over `ws` found -> dict(flagged_pairs, candidate_slots, slots, virtual_waypoints, flagged_fraction,
    evaluated_pairs: those the last culled pass 1 evaluated)."""
    st = torch.zeros(5, dtype=torch.int64, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_traj_step_stats(cloud.n, ws.n_virtual, ws.n_traj, ptr(ws.buf), ws.bytes, ptr(st), stream_ptr()), "tohip_traj_step_stats")
    f, c, s, v, e = (int(x) for x in st.cpu())
    return dict(flagged_pairs=f, candidate_slots=c, slots=s, virtual_waypoints=v, flagged_fraction=f / max(1, s * v), evaluated_pairs=e)


class PoseWorkspace:
    def __init__(self, cloud):
        self.bytes = _lib.lib().tohip_pose_workspace_bytes(cloud.n)
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=cloud.device)


def pose_forward(cloud, trans, quat, cam, ws, mask=None):
    obs = torch.empty(cloud.n, dtype=torch.float32, device=cloud.device)
    scalars = torch.zeros(4, dtype=torch.float32, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_pose_forward(ptr(cloud.blob), cloud.n, ptr(trans), ptr(quat), cam.ref(), ptr(mask),
                                            ptr(obs), ptr(scalars), ptr(ws.buf), ws.bytes, stream_ptr()),
              "tohip_pose_forward")
    return obs, scalars


def pose_forward_backward(cloud, trans, quat, cam, ws, mask=None, gout=None):
    """ModelPose.forward and the backward of its fused loss in ONE pass over the cloud (tohip_pose_forward_backward).
    -> (observations (N,), scalars (4: sum, loss, -, -), trans_grad (1,3), quat_grad (1,4)); gradients are gout x d loss / d (.)."""
    dev = cloud.device
    obs = torch.empty(cloud.n, dtype=torch.float32, device=dev)
    scalars = torch.empty(4, dtype=torch.float32, device=dev)
    tg = torch.empty((1, 3), dtype=torch.float32, device=dev)
    qg = torch.empty((1, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().tohip_pose_forward_backward(ptr(cloud.blob), cloud.n, ptr(trans), ptr(quat), cam.ref(), ptr(mask), ptr(obs), ptr(scalars),
                                                     ptr(gout), ptr(tg), ptr(qg), ptr(ws.buf), ws.bytes, stream_ptr()), "tohip_pose_forward_backward")
    return obs, scalars, tg, qg


def pose_backward(cloud, trans, quat, cam, ws, mask=None, grad_obs=None, scalars=None, gout=None):
    tg = torch.empty((1, 3), dtype=torch.float32, device=cloud.device)
    qg = torch.empty((1, 4), dtype=torch.float32, device=cloud.device)
    with torch.cuda.device(cloud.device):
        check(_lib.lib().tohip_pose_backward(ptr(cloud.blob), cloud.n, ptr(trans), ptr(quat), cam.ref(), ptr(mask),
                                             ptr(grad_obs), ptr(scalars), ptr(gout), ptr(tg), ptr(qg), ptr(ws.buf),
                                             ws.bytes, stream_ptr()), "tohip_pose_backward")
    return tg, qg


HPR_BATCH_POINTS = 32_000_000  # points per batched hull pass (workspace ~0.12 KB per point)


OCCLUSION_CULL_BYTES = 4 << 30  # budget for the cull stage's worst-case buffers (16 B per point and waypoint): waypoints are chunked


def occlusion_bits(cloud, points, poses, quats, cam, min_dist, max_dist, method="hpr"):
    """(W, npad/32) int32 occlusion bit rows for the given waypoints: the hard per-camera pipeline of
    /root/reference/src/pc_processor.py:158-187 (exact transform -> hard frustum cull -> HPR from the camera
    centre, or the z-buffer splat for method="zbuffer") turned into the bit layout the kernels read.
    The waypoints go through in chunks sized by OCCLUSION_CULL_BYTES (the cull stage sizes its outputs for the worst case)."""
    dev = cloud.device
    W = poses.shape[0]
    rows = torch.empty((W, cloud.npad // 32), dtype=torch.int32, device=dev)
    chunk = max(1, min(W, OCCLUSION_CULL_BYTES // (16 * max(cloud.n, 1))))
    for w0 in range(0, W, chunk):
        w1 = min(W, w0 + chunk)
        _occlusion_rows_chunk(cloud, points, poses[w0:w1].contiguous(), quats[w0:w1].contiguous(), cam, min_dist, max_dist, method,
                              rows[w0:w1])
    return rows


_SCRATCH = {}


def _scratch(device, tag, nbytes):
    """A grow-only byte buffer per (device, STREAM, purpose): the occlusion rows are rebuilt again and again over buffers of GBs
    whose sizes vary a little from call to call — allocating them anew every time cost more than the kernels (12 of 34 ms per
    rebuild at 1 M points x 128 waypoints).  Keyed by the current stream: two models rebuilding their masks on two streams do not
    share a buffer (on ONE stream the calls are ordered, and a buffer's content is dead when the call that filled it returns its
    rows).  A buffer that is outgrown is handed to the caching allocator with its stream recorded, so that work still queued on it
    finishes first.  release_scratch() gives the memory back."""
    stream = torch.cuda.current_stream(device)
    key = (str(device), stream.cuda_stream, tag)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        old = _SCRATCH.pop(key, None)
        if old is not None:
            old.record_stream(stream)
        t = _SCRATCH[key] = torch.empty(int(nbytes * 1.1) + 256, dtype=torch.uint8, device=device)
    return t


def release_scratch():
    _SCRATCH.clear()


def _occlusion_rows_chunk(cloud, points, poses, quats, cam, min_dist, max_dist, method, rows):
    L = _lib.lib()
    dev = cloud.device
    W = poses.shape[0]
    n = cloud.n
    # transform -> cull -> gather for all waypoints of the chunk in three launches (one host read: the counts size the hull pass);
    # for the hull pass the kept clouds are written end to end at once (r06: 128 device copies, 0.83 of a refresh's 11.1 ms, until then)
    packed = method != "zbuffer" and W <= 65535
    # ... and culled in the packed cloud's order (same points, same arithmetic per point; a hull is a property of the SET, and among
    # exact duplicates the lowest row is reported in either order: the packing sort is stable): kept indices = packed positions
    in_packed_order = packed and points.data_ptr() == cloud.points.data_ptr() and tuple(points.shape) == tuple(cloud.points.shape) and points.is_contiguous()
    if packed:
        kept_all, cat, counts, kcnt_all, seg_off_dev = cull_waypoints(_sorted_rows(cloud) if in_packed_order else points, poses, quats, cam, min_dist,
                                                                      max_dist, normalize=True, scratch=True, packed=True)
    else:
        kept_all, pts_all, counts, kcnt_all = cull_waypoints(points, poses, quats, cam, min_dist, max_dist, normalize=True, scratch=True)
    if method == "zbuffer":
        # every waypoint's z-buffer in the same three launches (chunks of as many z-buffers as 2 GB hold); visible[w, j] = 1 when
        # kept point j of waypoint w owns a pixel
        K9 = (ctypes.c_float * 9)(*[cam.c.K[i] for i in range(9)])
        width, height = int(cam.c.img_width), int(cam.c.img_height)
        wsb = L.tohip_zbuffer_batched_workspace_bytes(width, height, W)
        zws = _scratch(dev, "zbuf", wsb)
        visible = _scratch(dev, "zvis", 4 * W * max(n, 1))[:4 * W * max(n, 1)].view(torch.float32)
        seg_off = torch.arange(W, dtype=torch.int64, device=dev) * max(n, 1)
        with torch.cuda.device(dev):
            check(L.tohip_zbuffer_visible_batched(ptr(pts_all), max(n, 1), ptr(kcnt_all), W, K9, width, height, 0.03, float(min_dist), float(max_dist),
                                                  ptr(visible), ptr(zws), wsb, stream_ptr()), "tohip_zbuffer_visible_batched")
    else:
        # one batched hull pass over the waypoints' culled clouds laid end to end (several when they exceed HPR_BATCH_POINTS)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        tot = int(offs[-1])
        visible = _scratch(dev, "hvis", 4 * max(tot, 1))[:4 * max(tot, 1)].view(torch.float32)
        if not packed:
            cat = _scratch(dev, "hcat", 12 * max(tot, 1))[:12 * max(tot, 1)].view(torch.float32).view(-1, 3)
            for w in range(W):   # (device-to-device copies of each waypoint's kept rows: no host round trip)
                if counts[w]:
                    cat[int(offs[w]):int(offs[w + 1])].copy_(pts_all[w, :counts[w]])
        w0 = 0
        while w0 < W:
            w1 = w0 + 1
            while w1 < W and offs[w1 + 1] - offs[w0] <= HPR_BATCH_POINTS:
                w1 += 1
            lo, hi = int(offs[w0]), int(offs[w1])
            if hi > lo:
                status = _hpr_batched_mask(cat[lo:hi], [int(o) - lo for o in offs[w0:w1 + 1]], visible[lo:hi])
                st_h = status.cpu().numpy()   # (one host read for both checks)
                if (st_h == 3).any():
                    raise ValueError("Points cannot contain NaN")  # scipy's error in the reference's pipeline
                if ((st_h == 2) & (np.asarray(counts[w0:w1]) >= 4)).any():
                    raise _lib.HipError("occlusion_bits: a waypoint's culled cloud is flat (no 3-D hull; Qhull raises QH6154)")
            w0 = w1
        seg_off = seg_off_dev[:W] if packed else torch.from_numpy(offs[:W].copy()).to(dev)
    for w0 in range(0, W, 65535):
        w1 = min(W, w0 + 65535)
        with torch.cuda.device(dev):
            check(L.tohip_occlusion_rows_masked(n, None if in_packed_order else ptr(cloud.inv_perm), ptr(kept_all[w0:w1]), ptr(kcnt_all[w0:w1]), ptr(visible),
                                                ptr(seg_off[w0:w1]), 4, w1 - w0, ptr(rows[w0:w1]), stream_ptr()), "tohip_occlusion_rows_masked")


def _hpr_batched_mask(points, seg_offsets, mask_out):
    """tohip_hidden_pts_removal_batched for its mask only (mask_out: n_total f32, 1 = visible), over cached scratch -> status (B,) int32."""
    L = _lib.lib()
    n, dev = points.shape[0], points.device
    B = len(seg_offsets) - 1
    c_offs = (ctypes.c_int64 * (B + 1))(*[int(o) for o in seg_offsets])
    idx = _scratch(dev, "hidx", 4 * max(n, 1))[:4 * max(n, 1)].view(torch.int32)
    voff = torch.empty(B + 1, dtype=torch.int32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    wsb = L.tohip_hpr_batched_workspace_bytes(n, B)
    for _ in range(4):
        ws = _scratch(dev, "hull", wsb)
        with torch.cuda.device(dev):
            rc = L.tohip_hidden_pts_removal_batched(ptr(points), c_offs, B, 2.0, ptr(idx), ptr(voff), ptr(mask_out), ptr(status), ptr(ws), ws.numel(),
                                                    stream_ptr())
        if rc != _lib.ENOSPC:
            check(rc, "tohip_hidden_pts_removal_batched")
            return status
        wsb *= 4   # most of a cloud's points on its hull: every extra byte goes to faces
    raise _lib.HipError("hull workspace: still out of face capacity at 64x the recommended size")


def cull_waypoints(points, poses, quats, cam, min_dist, max_dist, normalize=True, scratch=False, packed=False):
    """Exact transform + hard frustum cull of `points` (N,3) for W poses at once (tohip_cull_waypoints).
    -> (kept_idx (W,N) int32, kept_pts (W,N,3) f32 camera frame, counts list[int], counts on the device (W,) int32): pose
    w's kept points are the first counts[w] rows of kept_idx[w] / kept_pts[w], in input order.  One host synchronisation
    (the counts).  scratch: the two worst-case sized outputs live in cached buffers (valid until the next such call).
    packed (tohip_cull_waypoints_packed; W <= 65535): kept_pts is (sum(counts), 3) instead — the poses' kept points end to end,
    pose w's in rows [offs[w], offs[w+1]) — and a fifth value, offs (W+1,) int64 on the device, is returned."""
    _require_cuda(points, "points")
    pts = points.detach().to(torch.float32).contiguous()
    dev, n, W = pts.device, pts.shape[0], poses.shape[0]
    p_in, q_in = poses.detach().to(torch.float32).contiguous(), quats.detach().to(torch.float32).contiguous()
    if packed and W > 65535:
        raise ValueError("cull_waypoints(packed=True) takes at most 65535 poses per call")
    if scratch:
        kept_all = _scratch(dev, "kept", 4 * W * max(n, 1))[:4 * W * max(n, 1)].view(torch.int32).view(W, max(n, 1))
        pts_all = _scratch(dev, "kpts", 12 * W * max(n, 1))[:12 * W * max(n, 1)].view(torch.float32).view(W, max(n, 1), 3)
    else:
        kept_all = torch.empty((W, max(n, 1)), dtype=torch.int32, device=dev)
        pts_all = torch.empty((W, max(n, 1), 3), dtype=torch.float32, device=dev)
    kcnt_all = torch.zeros(W, dtype=torch.int32, device=dev)
    L = _lib.lib()
    if packed:
        seg_off = torch.empty(W + 1, dtype=torch.int64, device=dev)
        wsb = L.tohip_cull_waypoints_workspace_bytes(n, W)
        fws = _scratch(dev, "cullws", wsb) if scratch else torch.empty(wsb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(L.tohip_cull_waypoints_packed(ptr(pts), n, ptr(p_in), ptr(q_in), W, int(bool(normalize)), cam.ref(), float(min_dist),
                                                float(max_dist), ptr(kept_all), ptr(pts_all), ptr(kcnt_all), ptr(seg_off), ptr(fws), wsb,
                                                stream_ptr()), "tohip_cull_waypoints_packed")
        counts = kcnt_all.cpu().tolist()
        return kept_all, pts_all.view(-1, 3)[:sum(counts)], counts, kcnt_all, seg_off
    for w0 in range(0, W, 65535):  # grid.y limit
        w1 = min(W, w0 + 65535)
        wsb = L.tohip_cull_waypoints_workspace_bytes(n, w1 - w0)
        fws = _scratch(dev, "cullws", wsb) if scratch else torch.empty(wsb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(L.tohip_cull_waypoints(ptr(pts), n, ptr(p_in[w0:w1]), ptr(q_in[w0:w1]), w1 - w0, int(bool(normalize)), cam.ref(),
                                         float(min_dist), float(max_dist), ptr(kept_all[w0:w1]), ptr(pts_all[w0:w1]),
                                         ptr(kcnt_all[w0:w1]), ptr(fws), wsb, stream_ptr()), "tohip_cull_waypoints")
    return kept_all, pts_all, kcnt_all.cpu().tolist(), kcnt_all


def to_camera_frame_exact(points, quat, trans, normalize=True, transpose=False):
    """Reference-exact f32 transform; (N,3) -> (N,3), or (3,N) when `transpose`."""
    _require_cuda(points, "points")
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    q = _dev_f32(quat, pts.device).reshape(4)
    t = _dev_f32(trans, pts.device).reshape(3)
    out = torch.empty((3, n) if transpose else (n, 3), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        check(_lib.lib().tohip_to_camera_frame(ptr(pts), n, ptr(q), ptr(t), int(normalize), int(transpose), ptr(out),
                                               stream_ptr()), "tohip_to_camera_frame")
    return out


def soft_masks(cam_points, cam, want_dist=True, want_fov=True):
    _require_cuda(cam_points, "points")
    pts = cam_points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    d = torch.empty(n, dtype=torch.float32, device=pts.device) if want_dist else None
    f = torch.empty(n, dtype=torch.float32, device=pts.device) if want_fov else None
    with torch.cuda.device(pts.device):
        check(_lib.lib().tohip_soft_masks(ptr(pts), n, cam.ref(), ptr(d), ptr(f), stream_ptr()), "tohip_soft_masks")
    return d, f


def soft_masks_backward(cam_points, cam, grad_dist=None, grad_fov=None):
    """dL/d cam_points (N,3) for upstream gradients of get_dist_mask and / or the soft get_fov_mask."""
    pts = cam_points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
    gd = grad_dist.to(torch.float32).contiguous() if grad_dist is not None else None
    gf = grad_fov.to(torch.float32).contiguous() if grad_fov is not None else None
    with torch.cuda.device(pts.device):
        check(_lib.lib().tohip_soft_masks_backward(ptr(pts), n, cam.ref(), ptr(gd), ptr(gf), ptr(out), stream_ptr()),
              "tohip_soft_masks_backward")
    return out


def to_camera_frame_backward(points, quat, trans, grad_out, want_points_grad=True):
    """-> (dL/d points (N,3) or None, dL/d quat (4,) raw quaternion, dL/d trans (3,))"""
    pts = points.detach().to(torch.float32).contiguous()
    n, dev = pts.shape[0], pts.device
    q = _dev_f32(quat, dev).reshape(4)
    t = _dev_f32(trans, dev).reshape(3)
    g = grad_out.to(torch.float32).contiguous()
    gx = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_points_grad else None
    gq = torch.empty(4, dtype=torch.float32, device=dev)
    gt = torch.empty(3, dtype=torch.float32, device=dev)
    L = _lib.lib()
    wsb = L.tohip_pose_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(L.tohip_to_camera_frame_backward(ptr(pts), n, ptr(q), ptr(t), ptr(g), ptr(gx), ptr(gq), ptr(gt), ptr(ws), wsb, stream_ptr()),
              "tohip_to_camera_frame_backward")
    return gx, gq, gt


def frustum_cull(cam_3xN, cam, min_dist, max_dist, want_indices=True):
    """-> (dist_mask bool[N], fov_mask bool[N], kept_idx int32[M] ascending)"""
    _require_cuda(cam_3xN, "points")
    pts = cam_3xN.detach().to(torch.float32).contiguous()
    n = pts.shape[1]
    dev = pts.device
    dm = torch.empty(n, dtype=torch.uint8, device=dev)
    fm = torch.empty(n, dtype=torch.uint8, device=dev)
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = _lib.lib().tohip_frustum_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().tohip_frustum_cull(ptr(pts), n, cam.ref(), float(min_dist), float(max_dist), ptr(dm), ptr(fm),
                                            ptr(idx) if want_indices else None, ptr(cnt), ptr(ws), wsb, stream_ptr()),
              "tohip_frustum_cull")
    m = int(cnt.item()) if want_indices else 0
    return dm.bool(), fm.bool(), idx[:m]


def spherical_flip(points, param=2):
    _require_cuda(points, "points")
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    out = torch.empty_like(pts)
    rad = torch.empty(1, dtype=torch.float32, device=pts.device)
    ws = torch.empty(8192, dtype=torch.uint8, device=pts.device)   # TOHIP_FLIP_WORKSPACE_BYTES
    with torch.cuda.device(pts.device):
        check(_lib.lib().tohip_spherical_flip(ptr(pts), n, float(param), ptr(out), ptr(rad), ptr(ws), 8192, stream_ptr()),
              "tohip_spherical_flip")
    return out, rad


def _with_hull_workspace(wsb, dev, call):
    """Run call(ws, wsb) with the recommended hull workspace; a cloud whose hull needs more faces than that holds
    (TOHIP_ENOSPC: most of its points are hull vertices) is retried with 4x the bytes — every extra byte goes to faces."""
    for _ in range(4):
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        try:
            return call(ws, wsb)
        except _lib.HipError as e:
            if e.code != _lib.ENOSPC:
                raise
            del ws
            wsb *= 4
    raise _lib.HipError("hull workspace: still out of face capacity at 64x the recommended size")


def hidden_pts_removal(points, param=2):
    """-> (visible_idx int32[V] ascending, mask f32[N])"""
    _require_cuda(points, "points")
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    dev = pts.device
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    mask = torch.empty(n, dtype=torch.float32, device=dev)
    def call(ws, wsb):
        with torch.cuda.device(dev):
            check(_lib.lib().tohip_hidden_pts_removal(ptr(pts), n, float(param), ptr(idx), ptr(cnt), ptr(mask), ptr(ws), wsb,
                                                      stream_ptr()), "tohip_hidden_pts_removal")
    _with_hull_workspace(_lib.lib().tohip_hpr_workspace_bytes(n), dev, call)
    return idx[:int(cnt.item())], mask


def hidden_pts_removal_batched(points, seg_offsets, param=2):
    """HPR of several independent clouds in one pass (one viewpoint = the origin of each): `points` (n_total,3)
    holds the segments end to end, `seg_offsets` (B+1 ints, host) their row ranges.
    -> (visible_idx int32 (rows of `points`, ascending), seg_visible_offsets int64 (B+1, host), mask f32[n_total],
        status int32[B] (0 ok, 1 fewer than 4 points, 2 flat, 3 NaN coordinates))"""
    _require_cuda(points, "points")
    pts = points.detach().to(torch.float32).contiguous()
    n, dev = pts.shape[0], pts.device
    offs = [int(o) for o in seg_offsets]
    B = len(offs) - 1
    if B < 1 or offs[0] != 0 or offs[-1] != n:
        raise ValueError("seg_offsets must run from 0 to len(points)")
    c_offs = (ctypes.c_int64 * (B + 1))(*offs)
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    voff = torch.empty(B + 1, dtype=torch.int32, device=dev)
    mask = torch.empty(n, dtype=torch.float32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    L = _lib.lib()
    def call(ws, wsb):
        with torch.cuda.device(dev):
            check(L.tohip_hidden_pts_removal_batched(ptr(pts), c_offs, B, float(param), ptr(idx), ptr(voff), ptr(mask),
                                                     ptr(status), ptr(ws), wsb, stream_ptr()), "tohip_hidden_pts_removal_batched")
    _with_hull_workspace(L.tohip_hpr_batched_workspace_bytes(n, B), dev, call)
    voff_h = voff.cpu().to(torch.int64)
    return idx[:int(voff_h[-1])], voff_h, mask, status


def hull_vertices_with_origin(points, with_origin=True, return_rounds=False):
    """Ascending hull-vertex indices of points (n,3) [+ the origin as index n] (tools.py:56-64)."""
    _require_cuda(points, "points")
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    dev = pts.device
    idx = torch.empty(n + 1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    rounds = ctypes.c_int32(0)
    def call(ws, wsb):
        with torch.cuda.device(dev):
            check(_lib.lib().tohip_convex_hull_vertices(ptr(pts), n, int(with_origin), ptr(idx), ptr(cnt),
                                                        ctypes.byref(rounds), ptr(ws), wsb, stream_ptr()),
                  "tohip_convex_hull_vertices")
    _with_hull_workspace(_lib.lib().tohip_hpr_workspace_bytes(n), dev, call)
    out = idx[:int(cnt.item())]
    return (out, rounds.value) if return_rounds else out


def render_points(verts, K, height, width, radius=0.03, znear=1.0, zfar=10.0, background=1.0, want_owner=False):
    """-> (image (H,W,3) f32, owner (H,W) int32 or None, owns_pixel (n,) bool)"""
    _require_cuda(verts, "verts")
    v = verts.detach().to(torch.float32).contiguous()
    n = v.shape[0]
    dev = v.device
    H, W = int(height), int(width)
    Kh = torch.as_tensor(K, dtype=torch.float32).detach().cpu()[:3, :3].reshape(9).tolist()
    Kc = (ctypes.c_float * 9)(*Kh)
    img = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    owner = torch.empty((H, W), dtype=torch.int32, device=dev) if want_owner else None
    owns = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    wsb = _lib.lib().tohip_render_workspace_bytes(W, H)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().tohip_render_points(ptr(v), n, Kc, W, H, float(radius), float(znear), float(zfar), float(background),
                                             ptr(img), ptr(owner), ptr(owns), ptr(ws), wsb, stream_ptr()), "tohip_render_points")
    return img, owner, owns[:n].bool()


def render_points_blend(verts, K, height, width, radius=0.03, znear=1.0, zfar=10.0, gamma=0.1, background=1.0):
    """-> image (H,W,3) f32: the depth-weighted blend of every disc over each pixel (render_kernels.hip; `gamma` = pulsar's softness)."""
    _require_cuda(verts, "verts")
    v = verts.detach().to(torch.float32).contiguous()
    dev = v.device
    H, W = int(height), int(width)
    Kh = torch.as_tensor(K, dtype=torch.float32).detach().cpu()[:3, :3].reshape(9).tolist()
    Kc = (ctypes.c_float * 9)(*Kh)
    img = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    wsb = _lib.lib().tohip_render_blend_workspace_bytes(W, H)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().tohip_render_points_blend(ptr(v), v.shape[0], Kc, W, H, float(radius), float(znear), float(zfar), float(gamma),
                                                   float(background), ptr(img), ptr(ws), wsb, stream_ptr()), "tohip_render_points_blend")
    return img


def selftest_wave_reduce(mat64xk):
    k = mat64xk.shape[1]
    dev = mat64xk.device
    s, mn, mx = (torch.empty(k, dtype=torch.float32, device=dev) for _ in range(3))
    with torch.cuda.device(dev):
        check(_lib.lib().tohip_selftest_wave_reduce(ptr(mat64xk.contiguous()), k, ptr(s), ptr(mn), ptr(mx), stream_ptr()),
              "tohip_selftest_wave_reduce")
    return s, mn, mx
