"""Sharding over the GPUs of one node (not in the reference, which is single-process): by waypoints
(WaypointShard, the north star's design) or by points (PointShard, the survey's alternative: collectives that do not grow
with the cloud).

WaypointShard.

One process per GPU; the cloud is replicated (12 MB at 1 M points), the evaluated waypoints are split
into contiguous per-rank ranges.  The only data-path collective is ONE all-reduce (sum, f32, N floats) of
the partial log-odds vector per forward — `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm;
the per-waypoint min/max normalisation is rank-local by construction.  In the backward every rank
produces the gradient rows of its own waypoints and a (W,7)-float all-reduce assembles them so that a
replicated optimiser steps identically everywhere.

PointShard.  Every rank packs N / R of the points and evaluates ALL waypoints on them.  A point's log-odds sum is then complete
on its own rank (rewards never travel); what the ranks exchange is per WAYPOINT: after pass 1 the extrema (one element-wise MAX
all-reduce of 4 int32 per virtual waypoint: the minimum is kept negated), after the gradient sums 40 doubles per virtual waypoint
plus the reward sum (one SUM all-reduce) — 2 KB and 41 KB at 128 waypoints whatever N is, against 4 MB per step for the log-odds
vector of a waypoint-sharded run at 1 M points.  Every rank ends up with the same loss and the same gradients.

The classes hold no kernels: they only place work and issue collectives, so their logic is covered by
world_size-2 gloo tests on CPU (tests/test_distributed_cpu.py).
"""
import torch
import torch.distributed as dist


class WaypointShard:
    kind = "waypoints"

    def __init__(self, process_group=None, force_collectives=False, compact=False):
        """force_collectives: issue the collectives even in a one-rank group (a rehearsal of the RCCL calls on one GPU).
        compact — EXPERIMENTAL, off by default, not used by any model or optimiser path: all-reduce only the 256-point slots some
        rank's forward listed as candidates (the log-odds vector is exactly zero elsewhere): a 0/1 flag per slot MAX-reduced first,
        then the union's slots summed — ~0.3 MB instead of 4 MB at 1 M points, for one host read of the union's size per step
        (ops.allreduce_log_odds).  Through a one-rank RCCL group it costs 65 us against 14.6 us for the full vector;
        it is kept only so that the first N > 1 run can measure both (bench.py's `comm.other_allreduce`) and is to be removed if the
        4 MB ring all-reduce stays below that there.  PointShard makes the question moot (its messages do not grow with the cloud)."""
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._always = bool(force_collectives)
        self.compact = bool(compact)

    def bounds(self, n_wps, rank=None):
        """Contiguous, balanced range [lo, hi) of the n_wps evaluated waypoints owned by `rank`."""
        r = self.rank if rank is None else rank
        base, rem = divmod(n_wps, self.world_size)
        lo = r * base + min(r, rem)
        return lo, lo + base + (1 if r < rem else 0)

    def allreduce_sum(self, t):
        return self._allreduce(t, dist.ReduceOp.SUM)

    def allreduce_max(self, t):
        """Element-wise maximum over the ranks, in place (the OR of 0/1 flags: RCCL has no bitwise reductions)."""
        return self._allreduce(t, dist.ReduceOp.MAX)

    def _allreduce(self, t, op):
        if self.world_size > 1 or self._always:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                # rehearsal setups (several ranks on one GPU, gloo): stage through the host; RCCL reduces in place
                h = t.detach().cpu()
                dist.all_reduce(h, op=op, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op, group=self.group)
        return t


    def allreduce_sum_async(self, t):
        """Starts the in-place sum of `t` and returns a handle whose wait() makes the CURRENT stream wait for it (RCCL
        runs the collective on its own stream): kernels enqueued in between that do not touch `t` overlap with it."""
        if (self.world_size > 1 or self._always) and not (t.is_cuda and dist.get_backend(self.group) == "gloo"):
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.allreduce_sum(t)  # single rank, or the host-staged rehearsal path: nothing left in flight

        class _Done:
            def wait(self):
                return True
        return _Done()

    def allgather_rows(self, t):
        """Concatenate every rank's (rows, k) block in rank order (equal row counts): the per-waypoint gradient
        rows of a contiguous shard -> the whole trajectory's, on every rank."""
        if self.world_size == 1 and not self._always:
            return t
        staged = t.is_cuda and dist.get_backend(self.group) == "gloo"
        src = t.detach().cpu() if staged else t.contiguous()
        out = src.new_empty((self.world_size * src.shape[0],) + tuple(src.shape[1:]))
        dist.all_gather_into_tensor(out, src, group=self.group)
        return out.to(t.device) if staged else out


class PointShard(WaypointShard):
    """Placement of the POINTS over the ranks: rank r owns the contiguous rows point_bounds(N) of the caller's cloud and every
    waypoint.  ModelTraj(points, ..., shard=PointShard()) takes the WHOLE cloud (or this rank's rows with n_points_global=) and
    keeps only its own rows; model.rewards are then this rank's rows' rewards."""
    kind = "points"

    def __init__(self, process_group=None, force_collectives=False):
        super().__init__(process_group, force_collectives, compact=False)

    def bounds(self, n_wps, rank=None):
        return 0, n_wps   # every rank evaluates every waypoint

    def point_bounds(self, n_points, rank=None):
        """Contiguous, balanced range [lo, hi) of the n_points rows owned by `rank`."""
        return WaypointShard.bounds(self, n_points, rank)


def init_from_env(backend=None, use_gpu=None, force=False):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun contract) and bind
    this process to its GPU (use_gpu=False: stay on the CPU, e.g. the gloo tests).  Returns (rank, world_size, device).
    force (or TOHIP_DIST_FORCE_INIT=1): create the process group even for one rank — the RCCL rehearsal on a single GPU."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available() if use_gpu is None else (use_gpu and torch.cuda.is_available())
    backend = backend or os.environ.get("TOHIP_DIST_BACKEND")  # "gloo": rehearse N ranks on fewer GPUs
    if use_gpu and backend == "gloo":
        local = local % torch.cuda.device_count()
    device = torch.device(f"cuda:{local}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    force = force or os.environ.get("TOHIP_DIST_FORCE_INIT") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")
        kw = {"device_id": device} if (use_gpu and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device
