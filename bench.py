"""bench.py — point-visibility evaluations/s (fwd+bwd) of the HIP hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W     -> the LAST stdout line is ONE compact JSON object (< 8 KB:
the driver keeps an 8 KB tail) on rank 0.  Everything else the run measures (side legs, per-kernel tables, the ISA mix) is printed
as EARLIER stdout lines ({"detail": <name>, ...}) and written to gpurun_out/bench_details.json (--details-file).
For N>1 the driver launches it under torch.distributed.run, one rank per GPU (RCCL); started plainly with --gpus N > 1 (no WORLD_SIZE
in the environment) it spawns that launcher itself, before anything touches the GPU, relays rank 0's line and exits with its status.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): a 1 M-point synthetic
cloud x 128 waypoints per GPU, full forward + backward to (x,y,z) and quaternion gradients, through the
C ABI of include/trajopt_hip.h.  One "step" at N = 1 = tohip_traj_forward_backward (five launches); with N > 1 a collective sits
between forward and backward: tohip_traj_forward -> all-reduce of the log-odds vector -> tohip_traj_reward_backward -> all-gather of the
(W,7) gradient rows.  With N
GPUs the trajectory has 128*N waypoints sharded contiguously over the ranks (weak scaling; N=8 is configs[3],
1 M x 1024); value = N_points * W_total / time.  Inputs are resident in HBM before the timed region.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed on the launch stream) and
"cpu_baseline" (the CPU oracle — a port, not the reference — on a bounded sample, rank 0, N=1 only).

The roofline is a VALU-ISSUE roofline: the kernels keep points in registers and stream waypoint records through SGPRs,
so a launch moves ~N*16 B whatever W is (1 % of HBM peak) and is bound by vector-instruction issue.
  achieved = issue cycles the launch's instruction stream needs / kernel time
             (instruction mix of the compiled inner loop: profiles/r05_pass1_isa_mix.json, from tools/isa_stats.py;
              prices per wave64 instruction measured on this chip: profiles/r02_valu_peak.json, tools/valu_peak.hip —
              packed f32 4, transcendental 8, other VALU 4 cycles)
  peak     = 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, max clock)
"""
import argparse
import gc
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from trajectory_optimization_amd import synth  # noqa: E402

N_POINTS = 1_000_000
WPS_PER_GPU = 128
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMDS = 1024          # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4         # max clock (the chip holds 2.1-2.4 GHz under this load: profiles/r02_valu_peak.json)
ISSUE_CYCLES = {"packed_f32": 4.0, "transcendental": 8.0, "other_valu": 4.0}  # per wave64 instruction, measured
# algorithmic bytes per evaluation of a STREAMING implementation of the reference's loop: kept as a named
# secondary figure only — the kernels do not stream, so it exceeds the HBM peak
ALGO_BYTES_FWD_BWD = 48.0
PASS1 = "k_traj_pass1"


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (name, parsed JSON), or (None, None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True):
        try:
            with open(path) as f:
                return os.path.basename(path), json.load(f)
        except (OSError, ValueError):
            continue
    return None, None


def source_hash():
    """sha256 of the sources the dense loop is compiled from (tools/isa_stats.py stores it with the mix it counted)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("common.hpp", "traj_kernels.hip"):
        with open(os.path.join(REPO, "trajectory_optimization_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def isa_mix():
    """VALU instructions of one (wave, waypoint) iteration of k_traj_pass1's dense inner loop (= 64*P evaluations), by class.
    The mix is a checked-in count of the compiled loop: it is only valid for the sources it was counted on."""
    name, d = _latest_profile("pass1_isa_mix.json")
    if d is None:
        raise SystemExit("profiles/rNN_pass1_isa_mix.json is missing (tools/isa_stats.py --json writes it)")
    d["stale"] = d.get("source_hash") != source_hash()
    d["file"] = "profiles/" + name
    return d


def pmc_figures(kernel):
    """Counter figures of `kernel` from the committed rocprofv3 --pmc passes of this same command (the latest
    profiles/rNN_bench_dense_pmc.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, in bytes; SQ_ACTIVE_INST_VALU x4 over the SIMD cycles of
    the kernel) -> dict(file, traffic, valu_busy, valu_busy_resident, kernel_ms) or {}.  Counters cannot be collected by the run that
    prints them (rocprofv3 wraps the process), so the line names the file and profile_age() compares its kernel time with the run's."""
    fname, d = _latest_profile("bench_dense_pmc.json")
    try:
        for name, v in d["kernels"].items():
            if kernel in name and "hbm_bytes_per_launch_corrected" in v:
                durs = [v[k] for k in v if k.startswith("duration_ns_mean_")]
                return {"file": "profiles/" + fname, "traffic": float(v["hbm_bytes_per_launch_corrected"]), "valu_busy": v.get("valu_busy_fraction"),
                        "valu_busy_resident": v.get("valu_busy_while_resident"), "kernel_ms": (min(durs) * 1e-6) if durs else None}
    except (TypeError, KeyError, ValueError):
        pass
    return {}


def profile_age(pmc, live_kernel_ms, tol=0.10):
    """The committed counters belong to a kernel that took pmc["kernel_ms"]; the run just measured live_kernel_ms for the same launch.
    More than `tol` apart -> the profile is from other sources or another clock regime: say so on the line (the HIP-event bracket of
    the live figure costs ~5 %, so the comparison allows for it)."""
    ref = pmc.get("kernel_ms")
    if not ref or not live_kernel_ms:
        return {"checked": False}
    ratio = live_kernel_ms / ref
    return {"checked": True, "profile_kernel_ms": ref, "live_kernel_ms": live_kernel_ms, "ratio": ratio,
            "stale": bool(abs(ratio - 1.0) > tol)}


def cpu_baseline(n_points, n_wps, n_wps_sample, timeout_s=240):
    """The oracle (oracle/vis_oracle.c, f32, OpenMP) on a bounded sample of the same workload — fwd+bwd over `n_wps_sample` of the
    waypoints — in a process of its own (oracle/cpu_baseline.py: OMP_NUM_THREADS = the cores this job may use (affinity and cgroup
    quota), OMP_PROC_BIND=close, OMP_PLACES=cores set before the OpenMP runtime loads; one warm-up, then the MEDIAN of >= 5 timed
    repetitions with every repetition's time on the line).  The child never touches the GPU."""
    import subprocess
    from oracle import cpu_baseline as cb
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--points", str(n_points), "--waypoints", str(n_wps), "--sample-waypoints", str(n_wps_sample)]
    r = subprocess.run(cmd, cwd=REPO, env=cb.child_env(), capture_output=True, text=True, timeout=timeout_s)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-300:], "kind": "port"}
    return json.loads(r.stdout.strip().splitlines()[-1])


def hpr_leg(points, device):
    """Hidden-point removal (flip + convex hull) of the same cloud seen from the origin: not bandwidth-characterisable,
    reported as time and hull points/s, next to scipy/Qhull — the hull the reference calls — on one host core."""
    from trajectory_optimization_amd import ops
    from oracle import oracle
    P = torch.from_numpy(points).to(device)
    ops.hidden_pts_removal(P)
    torch.cuda.synchronize(device)
    times = []
    for _ in range(8):   # the number of hull rounds (and with it the time) varies from build to build: mean and best of eight
        t0 = time.perf_counter()
        idx, _ = ops.hidden_pts_removal(P)
        torch.cuda.synchronize(device)
        times.append(time.perf_counter() - t0)
    gpu_s = sum(times) / len(times)
    t0 = time.perf_counter()
    ref, _ = oracle.hidden_pts_removal(points)
    cpu_s = time.perf_counter() - t0
    out = {"points": int(points.shape[0]), "visible": int(idx.numel()), "gpu_ms": 1e3 * gpu_s, "gpu_ms_best": 1e3 * min(times),
           "hull_points_per_s": points.shape[0] / gpu_s, "qhull_ms_host_1core": 1e3 * cpu_s,
           "index_set_equal_to_qhull": bool(np.array_equal(idx.cpu().numpy().astype(np.int64), ref))}
    # the hull as the occlusion refresh uses it: the cloud culled to the frustum of every waypoint of the bench path (camera frame,
    # viewpoint = the camera: pc_processor.py:158-187), all views in ONE batched pass; two of them checked against Qhull
    poses, quats = synth.make_path(WPS_PER_GPU, optical=True)
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    _, cat, counts, _, _ = ops.cull_waypoints(P, torch.from_numpy(poses).to(device), torch.from_numpy(quats).to(device), cam, 1.0, 15.0, packed=True)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ops.hidden_pts_removal_batched(cat, offs)
    torch.cuda.synchronize(device)
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        bidx, voff, _, status = ops.hidden_pts_removal_batched(cat, offs)
        torch.cuda.synchronize(device)
        times.append(time.perf_counter() - t0)
    bidx, voff = bidx.cpu().numpy().astype(np.int64), voff.numpy()
    same = True
    for w in (0, len(counts) // 2):
        if counts[w] >= 4:
            seg = cat[int(offs[w]):int(offs[w + 1])].cpu().numpy()
            same = same and bool(np.array_equal(bidx[voff[w]:voff[w + 1]] - offs[w], oracle.hidden_pts_removal(seg)[0]))
    out["batched"] = {"views": len(counts), "points": int(offs[-1]), "gpu_ms": 1e3 * sum(times) / len(times), "gpu_ms_best": 1e3 * min(times),
                      "status_ok": bool((status == 0).all().item()), "two_views_equal_to_qhull": same}
    return out


def dropin_leg(device, steps=200, warmup=10):
    """The reference's OWN loop (/root/reference/src/trajectory_optimization.py:109-116) over the drop-in classes:
        optimizer.zero_grad(); loss = model(); loss.backward(); optimizer.step()
    timed without any synchronisation inside the loop (one at the end), with torch.optim.Adam as the reference builds it (two
    parameter groups; nothing hooked: the hooks are opt-in since r06), the same optimizer after accelerate_torch_adam(opt) (its step()
    updates these Parameters with one launch), with torch's fused Adam, with this package's one-launch Adam (same constructor), with the whole step
    captured into a HIP graph (torch.cuda.graph + capturable Adam), and the launch-only optimize_trajectory beside them."""
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectory, accelerate_torch_adam, Adam as HipAdam
    b = np.load(os.path.join(REPO, "tests", "golden", "bundled.npz"))
    ident = np.tile(np.array([[1, 0, 0, 0]], np.float32), (len(b["poses"]), 1))
    cases = {"bundled_40k_x_27": (b["pts"].astype(np.float32), b["poses"].astype(np.float32), ident, 0.5),
             "synthetic_1M_x_128": (synth.make_cloud(N_POINTS, seed=0),) + synth.make_path(WPS_PER_GPU, optical=True) + (0.0,)}
    K = torch.from_numpy(synth.K_INTRINS)
    out = {}
    for name, (pts, poses, quats, vwd) in cases.items():
        P = torch.from_numpy(pts).to(device)

        def model():
            return ModelTraj(P, torch.from_numpy(poses), torch.from_numpy(quats), K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device)

        def groups(m):
            return [{"params": [m.poses], "lr": 0.1}, {"params": [m.quats], "lr": 0.02}]

        def loop(m, opt, n):
            for _ in range(n):
                opt.zero_grad()
                loss = m(vis_wps_dist=vwd)
                loss.backward()
                opt.step()
            return loss

        res = {}
        variants = {"torch.optim.Adam": lambda m: torch.optim.Adam(groups(m)),
                    "accelerate_torch_adam(torch.optim.Adam)": lambda m: accelerate_torch_adam(torch.optim.Adam(groups(m))),
                    "torch.optim.Adam(fused=True)": lambda m: torch.optim.Adam(groups(m), fused=True),
                    "trajectory_optimization_amd.optimizer.Adam": lambda m: HipAdam(groups(m))}
        for vname, mk in variants.items():
            try:
                m = model()
                opt = mk(m)
                loop(m, opt, warmup)
                torch.cuda.synchronize(device)
                gc_was = gc.isenabled()
                gc.disable()
                t0 = time.perf_counter()
                loss = loop(m, opt, steps)
                torch.cuda.synchronize(device)
                dt = time.perf_counter() - t0
                if gc_was:
                    gc.enable()
                res[vname] = {"ms_per_step": 1e3 * dt / steps, "loss_after": float(loss.item())}
            except Exception as e:  # a variant this torch build lacks is reported, not fatal
                res[vname] = {"error": f"{type(e).__name__}: {e}"[:200]}
        # the whole step (forward, backward, capturable Adam) captured once, replayed per step
        try:
            m = model()
            opt = torch.optim.Adam(groups(m), capturable=True)
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                loop(m, opt, 3)
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                loss = m(vis_wps_dist=vwd)
                loss.backward()
                opt.step()
            for _ in range(warmup):
                g.replay()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            res["hip_graph(torch.optim.Adam(capturable=True))"] = {"ms_per_step": 1e3 * dt / steps, "loss_after": float(loss.item())}
        except Exception as e:
            res["hip_graph(torch.optim.Adam(capturable=True))"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        m = model()
        optimize_trajectory(m, n_opt_steps=warmup, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=vwd)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        optimize_trajectory(m, n_opt_steps=steps, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=vwd)
        torch.cuda.synchronize(device)
        res["launch_only(optimize_trajectory)"] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / steps}
        n_eval = (len(poses) + m._wps_step(vwd) - 1) // m._wps_step(vwd)
        out[name] = {"points": int(pts.shape[0]), "waypoints": int(len(poses)), "waypoints_evaluated": int(n_eval), "steps": steps,
                     "mode": "culled (library default)", "variants": res}
    # the pose loop (/root/reference/src/pose_optimization.py:93-97,124-141) on the bundled cloud: ModelPose, two Adam groups
    from trajectory_optimization_amd.model import ModelPose
    from trajectory_optimization_amd.optimizer import optimize_pose
    Pb = torch.from_numpy(b["pts"].astype(np.float32)).to(device)

    def pose_model():
        return ModelPose(Pb, torch.tensor([[6.0, 2.0, 0.0]]), torch.tensor([[1.0, 0.0, 0.0, 0.0]]), K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device)
    res = {}
    for vname, mk in (("torch.optim.Adam", torch.optim.Adam), ("accelerate_torch_adam(torch.optim.Adam)", lambda g: accelerate_torch_adam(torch.optim.Adam(g))),
                      ("trajectory_optimization_amd.optimizer.Adam", HipAdam)):
        m = pose_model()
        opt = mk([{"params": [m.trans], "lr": 0.1}, {"params": [m.quat], "lr": 0.1}])

        def ploop(n):
            for _ in range(n):
                opt.zero_grad()
                loss = m()
                loss.backward()
                opt.step()
            return loss
        ploop(warmup)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        loss = ploop(steps)
        torch.cuda.synchronize(device)
        res[vname] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / steps, "loss_after": float(loss.item())}
    m = pose_model()
    optimize_pose(m, n_opt_steps=warmup)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    optimize_pose(m, n_opt_steps=steps)
    torch.cuda.synchronize(device)
    res["launch_only(optimize_pose)"] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / steps}
    out["pose_bundled_40k"] = {"points": int(Pb.shape[0]), "steps": steps, "variants": res}
    out["note"] = ("the reference's loop, unchanged, over ModelTraj: zero_grad(); loss = model(); loss.backward(); step() — host-bound: "
                   "model() and backward() are one library call each, the rest of the time is torch's autograd engine and optimizer")
    return out


def aux_leg(device, sizes=(1_000_000, 16_000_000), reps=20):
    """The HBM-bound kernels around the hot path (the only place BASELINE's "achieved HBM GB/s vs peak" literally
    applies): ModelPose's pass over the cloud (model.py:98-127), sphericalFlip (tools.py:38-53), the hard frustum cull
    (tools.py:176-187), the soft masks and to_camera_frame helpers (model.py:13-57), the PointCloud2 unpack
    (pointcloud_utils.py:197-198).  Per kernel family and cloud size: microseconds per call (HIP events on the launch stream around
    `reps` back-to-back calls through the C ABI, buffers allocated once), GB/s on the ALGORITHMIC bytes of SURVEY.md 8d (what a
    perfect streaming implementation must move), and that as a fraction of the 8 TB/s HBM peak."""
    from trajectory_optimization_amd import _lib, ops
    from trajectory_optimization_amd._lib import ptr, stream_ptr
    L = _lib.lib()
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    f32 = dict(dtype=torch.float32, device=device)
    out = {}
    for n in sizes:
        pts = torch.from_numpy(synth.make_cloud(n, seed=0)).to(device)
        cloud = ops.PackedCloud(pts, sort=False)           # ModelPose's layout: the caller's order
        ws = ops.PoseWorkspace(cloud)
        trans, quat = torch.tensor([[6.0, 2.0, 0.0]], **f32), torch.tensor([[0.9, 0.1, -0.3, 0.2]], **f32)
        obs, scal, tg, qg, gout = torch.empty(n, **f32), torch.zeros(4, **f32), torch.empty((1, 3), **f32), torch.empty((1, 4), **f32), torch.ones(1, **f32)
        cam3 = ops.to_camera_frame_exact(pts, quat, trans, normalize=True, transpose=True)    # (3, N) camera frame
        camn3 = cam3.t().contiguous()
        dm, fm = torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n, dtype=torch.uint8, device=device)
        kept, cnt = torch.empty(n, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int32, device=device)
        fws = torch.empty(L.tohip_frustum_workspace_bytes(n), dtype=torch.uint8, device=device)
        flipped, rad, flws = torch.empty_like(pts), torch.empty(1, **f32), torch.empty(8192, dtype=torch.uint8, device=device)
        d_m, f_m, out3 = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty((n, 3), **f32)
        msg = torch.zeros((n, 4), **f32)
        msg[:, :3] = pts
        raw = msg.view(torch.uint8).reshape(-1)
        msg_nan = msg.clone()   # the same message with 1 % of its rows invalid (a NaN in y): what a depth camera publishes
        msg_nan[torch.randperm(n, generator=torch.Generator().manual_seed(5))[: n // 100].to(device), 1] = float("nan")
        raw_nan = msg_nan.view(torch.uint8).reshape(-1)
        iws = torch.empty(L.tohip_ingest_workspace_bytes(n), dtype=torch.uint8, device=device)
        iws_nan = torch.empty(L.tohip_ingest_workspace_bytes(n), dtype=torch.uint8, device=device)
        q1, t1 = quat.reshape(4).contiguous(), trans.reshape(3).contiguous()
        s = stream_ptr()
        pblob = torch.empty(L.tohip_packed_cloud_bytes(n), dtype=torch.uint8, device=device)
        pws = torch.empty(L.tohip_pack_workspace_bytes(n), dtype=torch.uint8, device=device)
        vout, vws = torch.empty((n, 3), **f32), torch.empty(L.tohip_voxel_grid_workspace_bytes(n), dtype=torch.uint8, device=device)
        # the same points in an order that has locality (here: their own Morton order — what a map that was packed before, a voxel
        # grid's output or a lidar scan look like to the gathers; make_cloud's order is uniformly random, the gathers' worst case)
        _sorted = ops.PackedCloud(pts)
        pts_local = pts[_sorted.perm[:n].long()].contiguous()
        del _sorted
        calls = {
            "pack_cloud, Morton order (k_bbox + k_morton + radix passes + k_pack_cloud)": (32.0, lambda: L.tohip_pack_cloud(ptr(pts), n, 1, ptr(pblob), ptr(pws), pws.numel(), s)),
            "voxel_grid_filter, leaf 0.1 m (k_vox_bounds + keys + radix passes + heads + centroids)": (24.0, lambda: L.tohip_voxel_grid(ptr(pts), n, 0.1, 0.1, 0.1, 2, -2.5, 2.5, ptr(vout), ptr(cnt), ptr(vws), vws.numel(), s)),
            "pack_cloud, points arriving in an order with locality": (32.0, lambda: L.tohip_pack_cloud(ptr(pts_local), n, 1, ptr(pblob), ptr(pws), pws.numel(), s)),
            "voxel_grid_filter, points arriving in an order with locality": (24.0, lambda: L.tohip_voxel_grid(ptr(pts_local), n, 0.1, 0.1, 0.1, 2, -2.5, 2.5, ptr(vout), ptr(cnt), ptr(vws), vws.numel(), s)),
            "pose_forward (k_pose_stream<fwd>)": (16.0, lambda: L.tohip_pose_forward(ptr(cloud.blob), n, ptr(trans), ptr(quat), cam.ref(), None, ptr(obs), ptr(scal), ptr(ws.buf), ws.bytes, s)),
            "pose_forward_backward (k_pose_stream<fwd, grad>: one pass)": (16.0, lambda: L.tohip_pose_forward_backward(ptr(cloud.blob), n, ptr(trans), ptr(quat), cam.ref(), None, ptr(obs), ptr(scal), None, ptr(tg), ptr(qg), ptr(ws.buf), ws.bytes, s)),
            "pose_backward (k_pose_stream<grad>)": (12.0, lambda: L.tohip_pose_backward(ptr(cloud.blob), n, ptr(trans), ptr(quat), cam.ref(), None, None, ptr(scal), ptr(gout), ptr(tg), ptr(qg), ptr(ws.buf), ws.bytes, s)),
            "spherical_flip (k_norm_max + k_flip)": (24.0, lambda: L.tohip_spherical_flip(ptr(pts), n, 2.0, ptr(flipped), ptr(rad), ptr(flws), 8192, s)),
            "frustum_cull (k_frustum_count + k_frustum_write; above 2 M points a scan launch between them)": (None, lambda: L.tohip_frustum_cull(ptr(cam3), n, cam.ref(), 1.0, 10.0, ptr(dm), ptr(fm), ptr(kept), ptr(cnt), ptr(fws), fws.numel(), s)),
            "soft_masks (k_soft_masks)": (20.0, lambda: L.tohip_soft_masks(ptr(camn3), n, cam.ref(), ptr(d_m), ptr(f_m), s)),
            "to_camera_frame (k_to_camera_frame)": (24.0, lambda: L.tohip_to_camera_frame(ptr(pts), n, ptr(q1), ptr(t1), 1, 0, ptr(out3), s)),
            "pointcloud2_to_xyz (k_pc2_*: 16-byte xyzi points, no invalid row: one read of the message from the second call on)": (None, lambda: L.tohip_pointcloud2_to_xyz(ptr(raw), n, 16, 0, 4, 8, 7, 0, 1, ptr(out3), ptr(cnt), ptr(iws), iws.numel(), s)),
            "pointcloud2_to_xyz_1pct_nan_rows (k_pc2_*: the same message with 1 % invalid rows: count | scan | write)": (None, lambda: L.tohip_pointcloud2_to_xyz(ptr(raw_nan), n, 16, 0, 4, 8, 7, 0, 1, ptr(out3), ptr(cnt), ptr(iws_nan), iws_nan.numel(), s)),
        }
        rows = {}
        for name, (bpp, fn) in calls.items():
            for _ in range(3):
                rc = fn()
                if rc:
                    raise RuntimeError(f"{name}: error {rc}")
            def window(k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(k):
                    fn()
                e1.record()
                e1.synchronize()
                return 1e3 * e0.elapsed_time(e1) / k
            us = window(reps)
            if us * reps < 2000.0:   # a call of a few microseconds: a window of 20 is 0.2 ms, read on a clock ramp (r06: 11 - 16 us for the same
                us = min(window(max(reps, int(2500.0 / max(us, 1.0)))) for _ in range(3))   # kernels) -> the best of three windows of >= 2.5 ms
            if bpp is None:   # output bytes depend on what survives: the count of the last call
                k = int(cnt.item())
                bpp = (12.0 + 2.0 + 4.0 * k / n) if name.startswith("frustum") else (16.0 + 12.0 * k / n)
            gbs = bpp * n / (us * 1e-6) / 1e9
            rows[name] = {"us_per_call": us, "algorithmic_bytes_per_point": bpp, "GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
        out[f"{n}_points"] = rows
        del pts, pts_local, cloud, cam3, camn3, flipped, out3, msg, raw, msg_nan, raw_nan, obs, pblob, pws, vout, vws
        torch.cuda.empty_cache()
    out["note"] = ("algorithmic bytes per point: pack 12 B read + 20 B written, voxel grid 12 + (at most) 12 — both are bound by their sorts, not by these bytes; "
                   "pose 12 B read + 4 B written (16), backward alone 12; flip 12 + 12; cull 12 + 2 + 4 per kept point; "
                   "soft masks 12 + 8; to_camera_frame 12 + 12; PointCloud2 16 in + 12 per finite point.  HIP events over back-to-back calls: a "
                   "call's launches and their boundaries are part of its time (a 1 M-point call is mostly that).  Peak 8 TB/s; ~6.3 TB/s is "
                   "what streaming kernels achieve on this chip (MI355X_MICROARCH.md)")
    return out


def occlusion_leg(device, steps=20):
    """The occlusion-aware reward (the reference's TODO, tools.py:61-62 / model.py:210) on the headline workload:
    per waypoint the hard pipeline of pc_processor.py:171-178 (exact transform -> hard frustum cull -> HPR from the camera centre,
    or a z-buffer splat) turns into one bit per (waypoint, point).  Per method: milliseconds per mask refresh (all 128 waypoints),
    and per optimisation step (optimizer.optimize_trajectory) with the masks rebuilt every step and every tenth step
    (ModelTraj(occlusion_refresh_every=10): the masks are piecewise constant in the poses)."""
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    pts = torch.from_numpy(synth.make_cloud(N_POINTS, seed=0)).to(device)
    poses, quats = synth.make_path(WPS_PER_GPU, optical=True)
    K = torch.from_numpy(synth.K_INTRINS)
    out = {}
    for method in ("hpr", "zbuffer"):
        def model(k):
            return ModelTraj(pts, torch.from_numpy(poses), torch.from_numpy(quats), K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device,
                             occlusion=method, occlusion_refresh_every=k)
        m = model(1)
        ps, qs = m.poses.data, m.quats.data
        m._build_occlusion_rows(ps, qs)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(3):
            rows = m._build_occlusion_rows(ps, qs)
        torch.cuda.synchronize(device)
        refresh_ms = 1e3 * (time.perf_counter() - t0) / 3
        bits = rows.view(torch.int32)
        total_bits = 32.0 * bits.numel()
        set_bits = float(sum(int(((bits >> b) & 1).sum().item()) for b in range(32)))
        row = {"refresh_ms": refresh_ms, "hidden_fraction_of_pairs": 1.0 - set_bits / total_bits}
        for k in (1, 10):
            m = model(k)
            n = 5 if k == 1 else steps
            optimize_trajectory(m, n_opt_steps=1, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=0.0)
            m.refresh_occlusion()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            res = optimize_trajectory(m, n_opt_steps=n, lr_pose=0.1, lr_quat=0.02, rewards_th=1e9, vis_wps_dist=0.0)
            torch.cuda.synchronize(device)
            row[f"step_ms_refresh_every_{k}"] = 1e3 * (time.perf_counter() - t0) / n
            row[f"loss_after_{n}_steps_refresh_every_{k}"] = res.losses[-1]
        out[method] = row
        del m
    out["note"] = ("1 M points x 128 waypoints; refresh = cull of the cloud for every waypoint + one batched hull pass (hpr) or one z-buffer per "
                   "waypoint (zbuffer) + the bit rows; a step = optimize_trajectory's (forward | reward + backward | step tail) with the rows "
                   "multiplied into p")
    return out


def configs_leg(device, steps=20, warmup=5):
    """The other named sizes on ONE GPU (BASELINE.json configs 2, 4 and 5 — the N = 1 points of their curves — and eight concurrent
    trajectories): the loop the reference runs serially per waypoint (/root/reference/src/model.py:217-231), here with 1 024 - 1 280
    virtual waypoints in one launch sequence.  tools/prof_multi.py's scenarios, both evaluation modes: ms per step, evaluations/s,
    microseconds per kernel class (HIP events on the launch stream, a separate pass), what the forward found.
    profiles/r05_multi_*_kernel_stats.csv are the rocprofv3 summaries of the same steps."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import prof_multi
    out = {}
    for name, key in (("c2", "config2_100k_x_32_forward_only"), ("c3", "config3_1M_x_128"), ("w1024", "config4_work_on_one_gpu_1M_x_1024"),
                      ("cam5", "config5_work_on_one_gpu_5_cameras_x_1M_x_256"), ("multi8", "eight_trajectories_x_128_optimize_trajectories")):
        row = {}
        for mode in ("culled", "dense"):
            r = prof_multi.scenario(name, mode, device, steps, warmup)
            row.setdefault("what", r["what"])
            row[mode] = {k: r[k] for k in ("ms_per_step", "evals_per_s", "kernel_us", "flagged_pairs", "candidate_slots", "virtual_waypoints") if k in r}
            if mode == "culled":
                row[mode]["evaluated_pairs"] = r["evaluated_pairs_culled"]
            torch.cuda.empty_cache()
        out[key] = row
    out["note"] = ("one GPU, inputs resident; ms_per_step = wall over K steps between synchronisations after W warm-up steps; eight trajectories: "
                   "optimizer steps (tohip_traj_opt_step, the trajectories move while they are timed), the others tohip_traj_forward_backward "
                   "at fixed poses; kernel_us cost ~5 us per bracket: compare them with each other")
    return out


def message_leg(device, n_points=N_POINTS, n_wps=WPS_PER_GPU, opt_steps=30, reps=5):
    """One (cloud, path) message pair end to end, as the reference's TrajOpt.callback handles it
    (/root/reference/src/trajectory_optimization.py:60-81,129-157, behind launch/voxels_filtering.launch:11-21): PointCloud2 bytes
    (xyz + intensity, 16-byte points, 1 % NaN rows) -> host-to-device copy -> pointcloud2_to_xyz_array -> voxel_grid_filter (leaf
    0.1 m, z in [-2.5, 2.5]) -> ModelTraj (packs the cloud) -> opt_steps optimiser steps at the launch file's values -> poses back on
    the host.  Median of `reps` messages per stage, a synchronisation after each stage (total_ms: one message without them).
    The reference's comment for its loop alone: "~125 msec" per step (trajectory_optimization.py:108)."""
    from trajectory_optimization_amd import pointcloud_utils as pcu
    from trajectory_optimization_amd.model import ModelTraj
    from trajectory_optimization_amd.optimizer import optimize_trajectory
    rng = np.random.default_rng(7)
    pts = synth.make_cloud(n_points, seed=0)
    xyzi = np.concatenate([pts, rng.random((n_points, 1), dtype=np.float32)], axis=1)
    xyzi[rng.choice(n_points, n_points // 100, replace=False), rng.integers(0, 3, n_points // 100)] = np.nan
    msg = pcu.xyzi_array_to_pointcloud2(xyzi)
    poses, quats = synth.make_path(n_wps, optical=True)
    K = torch.from_numpy(synth.K_INTRINS)
    stages = {k: [] for k in ("pointcloud2_to_xyz_incl_h2d", "voxel_grid_filter", "model_and_pack", "optimize", "poses_to_host", "total_without_stage_syncs")}
    info = {}

    def once(sync):
        t = [time.perf_counter()]

        def mark():
            if sync:
                torch.cuda.synchronize(device)
            t.append(time.perf_counter())
        xyz = pcu.pointcloud2_to_xyz_array(msg, device=device)
        mark()
        vox = pcu.voxel_grid_filter(xyz, leaf_size=0.1)
        mark()
        m = ModelTraj(vox, torch.from_numpy(poses), torch.from_numpy(quats), K, synth.IMG_WIDTH, synth.IMG_HEIGHT, smoothness_weight=28.0, device=device)
        mark()
        r = optimize_trajectory(m, n_opt_steps=opt_steps, lr_pose=0.12, lr_quat=0.05, vis_wps_dist=0.0)
        mark()
        out = (m.poses.detach().cpu(), torch.nn.functional.normalize(m.quats.detach()).cpu())
        mark()
        info.update(points_in=n_points, points_finite=int(xyz.shape[0]), points_after_voxel_grid=int(vox.shape[0]), waypoints=n_wps,
                    optimiser_steps_taken=r.steps_taken, stopped_early=r.stopped)
        return [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])], out
    once(True)   # (allocator warm-up)
    for _ in range(reps):
        d, _ = once(True)
        for k, v in zip(list(stages)[:5], d):
            stages[k].append(v)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        once(False)
        torch.cuda.synchronize(device)
        stages["total_without_stage_syncs"].append(1e3 * (time.perf_counter() - t0))
    res = {k + "_ms": sorted(v)[len(v) // 2] for k, v in stages.items()}
    # the device part of the two format kernels alone (inputs resident): HIP events around back-to-back calls
    xyz = pcu.pointcloud2_to_xyz_array(msg, device=device)
    from trajectory_optimization_amd import ops
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.PackedCloud(xyz)
    e1.record()
    e1.synchronize()
    res["pack_cloud_device_ms"] = e0.elapsed_time(e1) / 20
    res.update(info)
    res["note"] = ("one message pair through the reference's callback path; pointcloud2_to_xyz includes the host-to-device copy of the 16 MB message "
                   "(PCIe) and the host read of the surviving count; optimize = device-resident loop (one host sync at its end); pack_cloud_device_ms: "
                   "bounding box, Morton keys, sort, gather, tile bounds of the finite points (rocprofv3: profiles/r05_message_kernel_stats.csv)")
    return res


def density_leg(device, steps=20, warmup=3):
    """The same 1 M points x 128 waypoints in ever smaller rooms (the path scaled with the room): the headline workload flags
    0.7 % of the (256-point slot, waypoint) pairs; an indoor cloud flags 10-20 %, and the kernels after pass 1 cost in proportion.
    Per extent: flagged fraction, dense and culled ms/step (tohip_traj_forward_backward, off the timed headline), bitwise equality
    of the two modes."""
    from trajectory_optimization_amd import ops
    out = []
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    gout = torch.ones(1, device=device)
    for ext in ((40.0, 40.0, 4.0), (20.0, 20.0, 4.0), (10.0, 10.0, 4.0), (6.0, 6.0, 3.0)):
        pts = synth.make_cloud(N_POINTS, seed=0, extent=ext)
        poses, quats = synth.make_path(WPS_PER_GPU, optical=True, scale=ext[0] / 40.0)
        cloud = ops.PackedCloud(torch.from_numpy(pts).to(device))
        p, q = torch.from_numpy(poses).to(device), torch.from_numpy(quats).to(device)
        ws = ops.TrajWorkspace(cloud, WPS_PER_GPU)
        row = {"extent_m": list(ext)}
        outs = {}
        for name, flags in (("dense", ops.DENSE), ("culled", 0)):
            for _ in range(warmup):
                o = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=flags)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                o = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=flags)
            torch.cuda.synchronize(device)
            row[f"{name}_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / steps
            outs[name] = o
        st = ops.traj_step_stats(cloud, ws)
        row.update(flagged_fraction=st["flagged_fraction"], flagged_pairs=st["flagged_pairs"], candidate_slots=st["candidate_slots"],
                   slots=st["slots"], mean_reward=float(outs["dense"][1][0].item()),
                   dense_equals_culled_bitwise=bool(all(torch.equal(a, b) for a, b in zip(outs["dense"][:5], outs["culled"][:5]))))
        out.append(row)
        del cloud, ws
    return out


def spawn_ranks(n, argv=None, script=None):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>`
    as a CHILD (never an exec: this pool forbids replacing a process, and this one must stay GPU-free), pass its stdout through with
    rank 0's headline held back to be the last line, return its exit status (non-zero if any rank failed: the launcher's status)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=REPO)
    headline = None
    for raw in proc.stdout:
        raw = raw.rstrip("\n")
        if raw.startswith('{"metric"'):
            headline = raw
        elif raw:
            print(raw, flush=True)
    rc = proc.wait()
    if headline is not None:
        print(headline, flush=True)
    elif rc == 0:
        print(f"bench.py: the {n} ranks exited 0 without a headline line", file=sys.stderr)
        rc = 1
    return rc


def _short(x, digits=6):
    """Floats to `digits` significant digits, recursively (the headline line is for reading and for an 8 KB tail)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _short(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, digits) for v in x]
    return x


def _drop_notes(x):
    return {k: _drop_notes(v) for k, v in x.items() if k != "note"} if isinstance(x, dict) else x


HEADLINE_BUDGET = 6000   # bytes; tests/test_host_cpu.py holds the line below 8 000 on canned numbers


def compact_line(full):
    """The ONE line the driver parses, cut from the full record: the contract's keys, a trimmed roofline, the culled default,
    cpu_baseline, sustained, and one or two figures of each side leg that ran.  Everything else stays in the details file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full[k] for k in keep if k in full}
    line["config"] = {k: v for k, v in full["config"].items() if k in ("workload", "n_points", "waypoints_total", "cameras", "parallelism", "launch", "mode", "loss_vis")}
    w = full.get("ms_per_step_windows") or {}
    line["ms_per_step_windows"] = {k: w[k] for k in ("median", "min", "max", "windows") if k in w}
    r = full["roofline"]
    line["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch", "kernel_share_of_step",
                                              "valu_busy_pmc", "valu_busy_while_resident_pmc", "hbm_counter_frac", "algorithmic_bytes_frac",
                                              "isa_mix_stale", "pmc_file", "profile_age")}
    ik = r.get("in_kernel")
    if ik:
        line["roofline"]["in_kernel_frac_of_peak"] = ik.get("frac_of_peak")
    line["roofline"]["note"] = ("VALU-issue roofline: achieved = ISA-counted issue cycles per launch / HIP-event kernel time (live); peak = 1024 SIMDs x 2.4 GHz; "
                                "traffic = HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) and valu_busy from the committed --pmc passes in pmc_file; "
                                "algorithmic_bytes_frac = 48 B/eval streaming model over 8 TB/s (> 1: points stay in registers, nothing streams)")
    c = full.get("culled_exact")
    if c:
        line["culled_exact"] = {k: c.get(k) for k in ("value", "unit", "ms_per_step", "bitwise_identical_to_dense")}
    if "cpu_baseline" in full:
        b = full["cpu_baseline"]
        line["cpu_baseline"] = {k: b[k] for k in ("value", "unit", "cores", "kind", "sample", "reps", "rep_s_min", "rep_s_max", "spread_max_over_min",
                                                  "nproc", "affinity", "cgroup_quota_cores", "error") if k in b}
    if full.get("sustained"):
        line["sustained"] = {m: {k: v[k] for k in ("ms_per_step", "seconds", "steps")} for m, v in full["sustained"].items() if isinstance(v, dict)}
        line["sustained"]["note"] = "back-to-back steps AFTER the timed windows (settled clocks); the headline is the driver's W + K protocol"
    if full.get("hpr"):
        line["hpr"] = {k: full["hpr"].get(k) for k in ("points", "visible", "gpu_ms", "gpu_ms_best", "qhull_ms_host_1core", "index_set_equal_to_qhull")}
        if isinstance(full["hpr"].get("batched"), dict):
            line["hpr"]["batched"] = {k: full["hpr"]["batched"].get(k) for k in ("views", "points", "gpu_ms", "status_ok", "two_views_equal_to_qhull")}
    if full.get("aux"):
        big = full["aux"].get("16000000_points") or next((v for k, v in full["aux"].items() if k.endswith("_points")), {})
        pick = {}
        for name, v in big.items():
            short = name.split(" ")[0].rstrip(",")
            if short in ("pointcloud2_to_xyz", "pointcloud2_to_xyz_1pct_nan_rows", "frustum_cull", "to_camera_frame", "pose_forward", "pose_forward_backward", "spherical_flip", "soft_masks") and short not in pick:
                pick[short] = {"us": v["us_per_call"], "frac_of_hbm_peak": v["frac_of_hbm_peak"]}
        line["aux_16M_points"] = pick
        small = full["aux"].get("1000000_points")
        if small:
            line["aux_1M_points"] = {name.split(" ")[0]: {"us": v["us_per_call"]} for name, v in small.items() if name.startswith("frustum_cull")}
    if full.get("after_optimisation"):
        m = full["after_optimisation"]
        line["after_optimisation"] = {"dense_ms_per_step": m["dense"]["ms_per_step"], "culled_ms_per_step": m["culled"]["ms_per_step"], "flagged_pairs": m.get("flagged_pairs")}
    if full.get("dropin"):
        d = full["dropin"].get("synthetic_1M_x_128", {}).get("variants", {})
        line["dropin_1M_x_128_ms_per_step"] = {k: v.get("ms_per_step") for k, v in d.items() if "ms_per_step" in v}
    if full.get("occlusion"):
        line["occlusion"] = {m: {k: v.get(k) for k in ("refresh_ms", "step_ms_refresh_every_10")} for m, v in full["occlusion"].items() if isinstance(v, dict)}
    if full.get("density_sweep"):
        line["density_sweep_culled_ms"] = {"x".join(str(int(e)) for e in row["extent_m"]): row["culled_ms_per_step"] for row in full["density_sweep"]}
    if full.get("configs"):
        line["configs_culled_ms_per_step"] = {k: v["culled"]["ms_per_step"] for k, v in full["configs"].items() if isinstance(v, dict) and "culled" in v}
    if full.get("message"):
        line["message_total_ms"] = full["message"].get("total_without_stage_syncs_ms")
    if full.get("comm"):
        cm = full["comm"]
        line["comm"] = _drop_notes(cm)
    for k in ("ranks_seen", "backend", "details_file", "bench_wall_s"):
        if k in full:
            line[k] = full[k]
    line = _short(line)
    out = json.dumps(line, separators=(",", ":"))
    if len(out) > HEADLINE_BUDGET:   # never let the line grow past the driver's tail again: drop the optional objects, largest first
        for k in sorted((k for k in line if k not in keep + ("config", "roofline", "cpu_baseline", "culled_exact", "sustained")),
                        key=lambda k: -len(json.dumps(line[k]))):
            line.pop(k)
            line["dropped_from_headline"] = line.get("dropped_from_headline", []) + [k]
            out = json.dumps(line, separators=(",", ":"))
            if len(out) <= HEADLINE_BUDGET:
                break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=100)   # (1.2 ms of warm-up leave the clocks still ramping: the first timed window then reads 3 % slow)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--wps-per-gpu", type=int, default=WPS_PER_GPU)
    ap.add_argument("--mode", choices=["both", "dense", "culled"], default="both",
                    help="dense = headline (every pair evaluated); culled = library default (exact skipping)")
    ap.add_argument("--cameras", type=int, default=1,
                    help="cameras per waypoint (BASELINE.json configs[4]: 5, with --wps-per-gpu 32); each (camera, "
                         "waypoint) pair is one virtual waypoint with its own min-max normalisation")
    ap.add_argument("--fused-reward", choices=["on", "off"], default="on",
                    help="on: tohip_traj_reward_backward (5 launches per step); off: tohip_traj_reward then tohip_traj_backward (6)")
    ap.add_argument("--fused-step", choices=["on", "off"], default="on",
                    help="N = 1: tohip_traj_forward_backward (5 launches); off: tohip_traj_forward then tohip_traj_reward_backward (5), the "
                         "split that a waypoint-sharded run needs around its all-reduce")
    ap.add_argument("--shard", choices=["waypoints", "points"], default="waypoints",
                    help="N > 1: what the ranks split.  waypoints (the north star's design): 128 waypoints per rank over the whole cloud, "
                         "one all-reduce of the N-float log-odds vector per step.  points: every rank takes points/N of the cloud and ALL "
                         "128 N waypoints (the same evaluations per rank); two collectives per step whose size does not depend on the cloud "
                         "(16 B per waypoint, then 320 B per waypoint)")
    ap.add_argument("--compact-allreduce", choices=["on", "off"], default="off",
                    help="EXPERIMENTAL. N > 1: all-reduce only the slots some rank lists as candidates (a flag per slot MAX-reduced first; one host read of "
                         "the union's size per step) instead of the whole N-float log-odds vector")
    ap.add_argument("--graph", choices=["on", "off"], default="off",
                    help="replay the step's launches from a HIP graph in the timed region (measured SLOWER on ROCm 7.2: 0.151 vs "
                         "0.139 ms dense, 0.079 vs 0.075 culled - graph kernel nodes cost more than the queue they replace)")
    ap.add_argument("--cpu-wps", type=int, default=64, help="waypoints in the CPU-baseline sample (0 = skip the N = 1 side legs altogether)")
    ap.add_argument("--details", choices=["off", "brief", "full"], default="brief",
                    help="side legs beside the headline (N = 1 only): off = none; brief (default, what the driver's command runs) = sustained, "
                         "hpr, aux at 16 M points, cpu_baseline; full = also dropin, configs, message, density, occlusion, moved, aux at 1 M.  "
                         "Each leg's own flag overrides the level")
    ap.add_argument("--details-file", default=os.path.join(REPO, "gpurun_out", "bench_details.json"),
                    help="where the full record (headline + every leg + per-kernel tables + ISA mix) is written; 'none' = nowhere")
    for leg, what in (("dropin", "time the reference's own loop over the drop-in classes"),
                      ("configs", "the other named sizes on one GPU: configs 2, 4, 5 and eight concurrent trajectories"),
                      ("message", "one PointCloud2 + path message pair end to end, per stage"),
                      ("density", "step time versus flagged fraction: 1 M points in ever smaller rooms"),
                      ("occlusion", "the occlusion-aware reward on the headline workload: ms per mask refresh and per step"),
                      ("aux", "the HBM-bound kernels around the hot path (ModelPose, flip, cull, masks, ingest)"),
                      ("sustained", "seconds of back-to-back steps per mode AFTER the timed windows: the step at settled clocks"),
                      ("moved", "time the same step on the trajectory after 100 optimiser steps"),
                      ("hpr", "hidden-point removal of the cloud from the origin: ms, hull points/s, index set against Qhull"),
                      ("cpu", "the CPU oracle on a bounded sample (cpu_baseline)")):
        ap.add_argument(f"--{leg}", choices=["on", "off"], default=None, help=what + " (N = 1 only)")
    ap.add_argument("--dump", default=None, help="write the last dense step's outputs (scalars, gradient rows, rewards) to this .npz "
                                                 "(rank 0): tests compare runs at different N")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: this process has not touched the GPU (importing torch does not) and never will — it starts the ranks as
        # children, relays their output and exits with their status
        raise SystemExit(spawn_ranks(args.gpus))
    BRIEF, FULL = {"sustained", "hpr", "aux", "cpu"}, {"sustained", "hpr", "aux", "cpu", "dropin", "configs", "message", "density", "occlusion", "moved"}

    def leg_on(name):
        v = getattr(args, name)
        if v is not None:
            return v == "on"
        return name in {"off": set(), "brief": BRIEF, "full": FULL}[args.details]

    from trajectory_optimization_amd import _lib, ops
    from trajectory_optimization_amd.distributed import init_from_env, WaypointShard, PointShard
    import torch.distributed as dist

    rank, world, device = init_from_env()
    if world != args.gpus and args.gpus != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the visibility path has no CPU fallback")
    n_gpus = world
    w_total = args.wps_per_gpu * n_gpus

    # ---- synthetic inputs (seeded), resident in HBM ------------------------------------
    pts = synth.make_cloud(args.points, seed=0)
    poses_all, quats_all = synth.make_path(w_total, optical=True)
    forced = os.environ.get("TOHIP_DIST_FORCE_INIT") == "1"   # one-rank process group: the RCCL calls of the N>1 step on one GPU
    by_points = args.shard == "points" and (n_gpus > 1 or forced)
    if by_points:
        # this rank's rows of the cloud, every waypoint of the whole trajectory: the same evaluations per rank as a waypoint shard
        shard = PointShard(force_collectives=forced)
        p_lo, p_hi = shard.point_bounds(args.points)
        lo, hi = 0, w_total
        cloud = ops.PackedCloud(torch.from_numpy(pts[p_lo:p_hi].copy()).to(device))
    else:
        lo, hi = rank * args.wps_per_gpu, (rank + 1) * args.wps_per_gpu
        cloud = ops.PackedCloud(torch.from_numpy(pts).to(device))
        shard = WaypointShard(force_collectives=forced, compact=args.compact_allreduce == "on") if (n_gpus > 1 or forced) else None
    cam = ops.Camera(synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT)
    poses = torch.from_numpy(poses_all[lo:hi].copy()).to(device)
    quats = torch.from_numpy(quats_all[lo:hi].copy()).to(device)
    n_local_wps = hi - lo
    n_virtual = n_local_wps * args.cameras
    rig = ops.CameraRig(*synth.camera_rig(args.cameras), device) if args.cameras > 1 else None
    ws = ops.TrajWorkspace(cloud, n_virtual)
    gout = torch.ones(1, device=device)
    pstep = {}   # --shard points: one ops.PointShardStep per evaluation mode (its buffers are allocated once)

    if shard is not None:
        # communicator set-up (RCCL rings over xGMI) happens on the first collective of each kind: keep it out of the timed
        # region whatever --warmup says
        if by_points:
            shard.allreduce_max(torch.zeros(4 * n_virtual, dtype=torch.int32, device=device))
            shard.allreduce_sum(torch.zeros(8 + 40 * n_virtual, dtype=torch.float64, device=device))
        else:
            shard.allreduce_sum(torch.zeros(cloud.npad, device=device))
            shard.allgather_rows(torch.zeros((args.wps_per_gpu, 7), device=device))
        torch.cuda.synchronize(device)

    rewards_buf = torch.empty(cloud.n, dtype=torch.float32, device=device)   # refilled by every forward
    lo_buf = torch.empty(cloud.npad, dtype=torch.float32, device=device)
    mm_buf = torch.empty((n_virtual, 2), dtype=torch.float32, device=device)

    at = {"poses": poses, "quats": quats}   # where the timed steps evaluate (moved_leg swaps in the trajectory after an optimisation run)

    def step(flags):
        rewards = rewards_buf
        poses, quats = at["poses"], at["quats"]
        if by_points:
            # pass 1 | MAX of the waypoints' extrema | flags, log-odds, rewards, gradient sums | SUM of 40 doubles per waypoint | finish
            st = pstep.get(flags)
            if st is None:
                st = pstep[flags] = ops.PointShardStep(cloud, args.points, n_local_wps, cam, ws, shard, rig=rig, flags=flags)
            rewards, scalars, pg, qg = st.step(poses, quats)
            return scalars, pg, qg, rewards
        if shard is None and args.fused_step == "on":
            # no collective between forward and backward: the whole step is ONE library call, five launches
            rewards, scalars, pg, qg, _, _ = ops.traj_forward_backward(cloud, poses, quats, cam, ws, gout, rig=rig, flags=flags, lo_sum=lo_buf,
                                                                       minmax=mm_buf, rewards=rewards)
            return scalars, pg, qg, rewards
        lo_sum, minmax = ops.traj_forward(cloud, poses, quats, cam, ws, rig=rig, flags=flags, lo_sum=lo_buf, minmax=mm_buf, rewards_half=rewards)
        if shard is not None:
            ops.allreduce_log_odds(shard, cloud, ws, lo_sum)  # the one data-path collective: N floats over xGMI (or the union's slots)
        if args.fused_reward == "on":
            # rewards, mean and loss share the backward's first launch (tohip_traj_reward_backward: two launches instead of three)
            rewards, scalars, pg, qg = ops.traj_reward_backward(cloud, n_local_wps, cam, ws, lo_sum, gout, rewards=rewards,
                                                               prefilled=True, rig=rig, flags=flags)
        else:
            rewards, scalars = ops.traj_reward(cloud, lo_sum, cam, ws, rewards=rewards, prefilled=True)
            pg, qg = ops.traj_backward(cloud, n_local_wps, cam, ws, lo_sum, scalars=scalars, gout=gout, rig=rig, flags=flags)
        if shard is not None:
            g = shard.allgather_rows(torch.cat([pg, qg], dim=1))  # (W_total, 7) floats: every rank can step the optimiser
            pg, qg = g[:, :3], g[:, 3:]
        return scalars, pg, qg, rewards

    def barrier():
        # under RCCL the barrier is an all-reduce on a device: name it (else torch guesses it from the rank and warns)
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()

    def fence():
        if n_gpus > 1 or forced:
            barrier()
        torch.cuda.synchronize(device)

    L = _lib.lib()
    ms = (ctypes.c_double * 6)()
    cnt = (ctypes.c_int64 * 6)()

    use_graph = args.graph == "on" and shard is None

    def captured(flags):
        """The step's launches captured once into a HIP graph (torch.cuda.CUDAGraph on ROCm): a replay enqueues the same
        kernels with the same arguments from one host call, so the ~50 us of Python + ctypes per step cannot starve the queue."""
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(2):
                step(flags)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o = step(flags)
        return g, o

    def timed(flags):
        """W warm-up steps, then exactly K steps between two (barrier + synchronize) fences; MAX over ranks.
        No instrumentation inside: the library's per-kernel HIP events cost ~0.06 ms per step."""
        if use_graph:
            g, o = captured(flags)
            run = g.replay
        else:
            o = None
            run = None
        for _ in range(args.warmup):
            run() if run else step(flags)
        fence()
        gc_was = gc.isenabled()
        gc.disable()   # a collector pause in the issuing thread is not part of the path
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if run:
                run()
            else:
                o = step(flags)
        fence()
        dt = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        if n_gpus > 1 or forced:
            t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, o

    def windows(flags, n_windows=5):
        """ms per step of n further windows of K steps each (same fences): the spread of the one timed window."""
        out = []
        for _ in range(n_windows):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(flags)
            fence()
            out.append(1e3 * (time.perf_counter() - t0) / args.steps)
        return sorted(out)

    def sustained_leg(seconds=2.0, chunk=500):
        """Back-to-back steps for `seconds` per mode (no synchronisation inside a chunk of `chunk` steps): what a step costs after
        seconds at load — the clocks and the temperature have settled — beside the K-step window the headline is timed on."""
        res = {}
        for name, flags in (("culled", 0), ("dense", dense_flags)):
            fence()
            t_start = time.perf_counter()
            chunks = []
            while time.perf_counter() - t_start < seconds:
                t0 = time.perf_counter()
                for _ in range(chunk):
                    step(flags)
                torch.cuda.synchronize(device)
                chunks.append(1e3 * (time.perf_counter() - t0) / chunk)
            res[name] = {"seconds": time.perf_counter() - t_start, "steps": chunk * len(chunks), "ms_per_step": sum(chunks) / len(chunks),
                         "ms_per_step_first_chunk": chunks[0], "ms_per_step_last_chunk": chunks[-1], "ms_per_step_best_chunk": min(chunks)}
        res["note"] = f"chunks of {chunk} steps, one synchronisation per chunk; the culled step is close to what one host thread can issue (~0.045 ms per call)"
        return res

    def comm_leg_points(flags):
        """--shard points: the two collectives alone (events on the compute stream around K of each, which waits for RCCL's), and
        the step's launches with the collectives left out."""
        st = pstep[flags]
        out = {}
        for name, t, fn in (("extrema_allreduce_max", st.extrema, shard.allreduce_max), ("sums_allreduce_sum", st.partial, shard.allreduce_sum)):
            keep = t.clone()
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
            fence()
            for e in ev:
                t.copy_(keep)
                e[0].record()
                fn(t)
                e[1].record()
            fence()
            d = sorted(e[0].elapsed_time(e[1]) for e in ev)
            out[name] = {"ms_median": d[len(d) // 2], "ms_max": d[-1], "bytes": int(t.numel() * t.element_size())}
            t.copy_(keep)

        class _NoComm:
            world_size, rank, kind = shard.world_size, shard.rank, "points"

            @staticmethod
            def allreduce_max(t):
                return t

            @staticmethod
            def allreduce_sum(t):
                return t
        real, st.shard = st.shard, _NoComm
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.step(at["poses"], at["quats"])
        fence()
        st.shard = real
        out.update(step_without_collectives_ms=1e3 * (time.perf_counter() - t0) / args.steps, backend=dist.get_backend(), shard="points",
                   note="rank 0's view; the collectives' sizes do not depend on the cloud: 16 B and 320 B per virtual waypoint (+ 64 B)")
        return out

    def comm_leg(flags):
        """N > 1: where the step's time goes.  The same K steps with events around the two collectives (on the compute stream,
        which waits for RCCL's), and K steps of the same launches with the collectives left out."""
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
        fence()
        for e in ev:
            lo_sum, _ = ops.traj_forward(cloud, poses, quats, cam, ws, rig=rig, flags=flags, lo_sum=lo_buf, minmax=mm_buf, rewards_half=rewards_buf)
            e[0].record()
            ops.allreduce_log_odds(shard, cloud, ws, lo_sum)
            e[1].record()
            _, _, pg, qg = ops.traj_reward_backward(cloud, args.wps_per_gpu, cam, ws, lo_sum, gout, rewards=rewards_buf, prefilled=True, rig=rig,
                                                    flags=flags)
            g = torch.cat([pg, qg], dim=1)
            e[2].record()
            shard.allgather_rows(g)
            e[3].record()
        fence()
        ar = sorted(e[0].elapsed_time(e[1]) for e in ev)
        ag = sorted(e[2].elapsed_time(e[3]) for e in ev)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lo_sum, _ = ops.traj_forward(cloud, poses, quats, cam, ws, rig=rig, flags=flags, lo_sum=lo_buf, minmax=mm_buf, rewards_half=rewards_buf)
            ops.traj_reward_backward(cloud, args.wps_per_gpu, cam, ws, lo_sum, gout, rewards=rewards_buf, prefilled=True, rig=rig, flags=flags)
        fence()
        no_comm = 1e3 * (time.perf_counter() - t0) / args.steps
        # the other kind of all-reduce of the log-odds vector (full N floats <-> the union of the ranks' candidate slots), for comparison
        other = WaypointShard(force_collectives=forced, compact=not shard.compact)
        ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
        fence()
        for e in ev2:
            lo_sum, _ = ops.traj_forward(cloud, poses, quats, cam, ws, rig=rig, flags=flags, lo_sum=lo_buf, minmax=mm_buf, rewards_half=rewards_buf)
            e[0].record()
            ops.allreduce_log_odds(other, cloud, ws, lo_sum)
            e[1].record()
        fence()
        ar2 = sorted(e[0].elapsed_time(e[1]) for e in ev2)
        return {"allreduce_ms_median": ar[len(ar) // 2], "allreduce_ms_max": ar[-1], "allreduce_bytes": int(cloud.npad * 4), "compact_allreduce": bool(shard.compact),
                "other_allreduce": {"compact": bool(other.compact), "ms_median": ar2[len(ar2) // 2], "ms_max": ar2[-1],
                                    "note": "the same K forwards followed by the OTHER kind of all-reduce (compact: a 0/1 flag per slot MAX-reduced, one "
                                            "host read of the union's size, pack, sum of the union's slots, unpack), events around it"},
                "allgather_ms_median": ag[len(ag) // 2], "allgather_bytes_per_rank": int(args.wps_per_gpu * 7 * 4),
                "step_without_collectives_ms": no_comm, "backend": dist.get_backend(),
                "note": "rank 0's view; events on the compute stream around each collective (it waits for RCCL's stream), "
                        "and the same launches with the collectives left out"}

    def kernel_times(flags):
        """The same K steps once more with HIP events recorded around every kernel on the launch stream
        (tohip_profile_*): mean duration per launch, for the roofline object."""
        fence()
        L.tohip_profile_enable(1)
        for _ in range(args.steps):
            step(flags)
        fence()
        _lib.check(L.tohip_profile_read(ms, cnt), "tohip_profile_read")
        L.tohip_profile_enable(0)
        return {L.tohip_profile_name(i).decode(): (ms[i], cnt[i]) for i in range(6) if cnt[i] > 0}

    def moved_leg(opt_steps=100, lr=(0.1, 0.02)):
        """The SAME step where an optimisation run ends up instead of where it starts: the trajectory after `opt_steps` steps of
        optimizer.optimize_trajectory (Adam, lr_pose / lr_quat as the drop-in leg, no early stop).  Rewards improve as the
        waypoints spread out, so more (slot, waypoint) pairs are flagged in more candidate slots and the kernels behind pass 1
        cost more: K steps of each mode timed like the headline (off the headline), per-kernel HIP-event times, and what the
        forward found."""
        from trajectory_optimization_amd.model import ModelTraj
        from trajectory_optimization_amd.optimizer import optimize_trajectory
        m = ModelTraj(torch.from_numpy(pts).to(device), torch.from_numpy(poses_all[lo:hi].copy()), torch.from_numpy(quats_all[lo:hi].copy()),
                      torch.from_numpy(synth.K_INTRINS), synth.IMG_WIDTH, synth.IMG_HEIGHT, device=device)
        optimize_trajectory(m, n_opt_steps=opt_steps, lr_pose=lr[0], lr_quat=lr[1], rewards_th=1e9, vis_wps_dist=0.0)
        moved_p, moved_q = m.poses.data.clone(), m.quats.data.clone()
        disp = (moved_p - poses).norm(dim=1)
        res = {"optimiser_steps": opt_steps, "lr_pose": lr[0], "lr_quat": lr[1], "mean_waypoint_displacement_m": float(disp.mean().item()),
               "max_waypoint_displacement_m": float(disp.max().item())}
        del m
        at["poses"], at["quats"] = moved_p, moved_q
        try:
            outs = {}
            for name, flags in (("dense", ops.DENSE), ("culled", 0)):
                k = kernel_times(flags)
                dt_m, o = timed(flags)
                outs[name] = tuple(t.clone() for t in o)
                res[name] = {"ms_per_step": 1e3 * dt_m / args.steps, "kernel_us": {kk: 1e3 * v[0] / v[1] for kk, v in k.items()}}
            st = ops.traj_step_stats(cloud, ws)
            res.update(flagged_pairs=st["flagged_pairs"], flagged_fraction=st["flagged_fraction"], candidate_slots=st["candidate_slots"],
                       slots=st["slots"], evaluated_pairs_culled=st["evaluated_pairs"], loss_vis=float(outs["dense"][0][1].item()),
                       dense_equals_culled_bitwise=bool(all(torch.equal(a, b) for a, b in zip(outs["dense"], outs["culled"]))))
        finally:
            at["poses"], at["quats"] = poses, quats
        res["note"] = ("the headline's step on the trajectory an optimisation run leaves behind; kernel_us: HIP events on the launch stream "
                       "(they cost ~5 us per bracket: compare them with each other, ms_per_step is timed without them)")
        return res

    def in_kernel_span(flags):
        """Diagnostic, outside every timed region: the dense kernel's blocks stamp s_memrealtime (100 MHz) at their start and end
        (tohip_profile_clock); first start -> last end per XCD (each XCD has its own counter), the longest of the eight.
        rocprofv3 / HIP-event durations of back-to-back launches also contain the queueing behind the previous kernel."""
        nb = L.tohip_profile_clock_blocks(cloud.n, n_virtual, flags, 0)
        if nb <= 0:
            return None, None
        buf = torch.zeros(6 * nb, dtype=torch.int64, device=device)
        L.tohip_profile_clock(buf.data_ptr())
        for _ in range(3):
            step(flags)
        torch.cuda.synchronize(device)
        L.tohip_profile_clock(None)
        raw = buf.cpu().numpy()
        pair, ext = raw[:2 * nb].reshape(nb, 2).astype(np.float64), raw[2 * nb:].reshape(nb, 4)
        ok = pair[:, 1] > 0
        spans = [(ext[(ext[:, 3] & 0xf) == x, 1].max() - ext[(ext[:, 3] & 0xf) == x, 0].min()) * 1e-5
                 for x in range(8) if ((ext[:, 3] & 0xf) == x).any()]
        return max(spans), float(np.median(pair[ok, 0] / pair[ok, 1]) * 0.1)

    # headline: DENSE — every (point, waypoint) pair is evaluated.  The instrumented pass (K steps with HIP events: the roofline's
    # kernel time) runs first, then the driver's protocol exactly: W warm-up steps + K timed steps between fences.  Nothing else is
    # run ahead of the timed window (r05 ran 2 x 2 s of load first, which moved the figure by clocks, not by kernels); the settled-clock
    # figure is the `sustained` object, measured AFTER the windows.
    t_bench0 = time.perf_counter()
    dense_flags = ops.DENSE if args.mode != "culled" else 0
    kern = kernel_times(dense_flags)
    dt, out = timed(dense_flags)
    out = tuple(t.clone() for t in out)   # the step's outputs live in buffers the later legs reuse
    span_ms, clock_ghz = in_kernel_span(dense_flags) if (dense_flags and n_gpus == 1) else (None, None)
    # the library's default path: exact culling (bitwise identical outputs, tests/test_hip_traj.py)
    if args.mode == "both":
        kern_c = kernel_times(0)
        dt_c, out_c = timed(0)
        out_c = tuple(t.clone() for t in out_c)
    else:
        dt_c, out_c, kern_c = dt, out, kern
    evals_per_step = args.points * w_total * args.cameras
    value = evals_per_step * args.steps / dt
    win_dense = windows(dense_flags)
    win_culled = windows(0) if args.mode == "both" else win_dense
    side = shard is None and n_gpus == 1 and args.cpu_wps > 0 and args.cameras == 1   # the N = 1 side legs
    sustained = sustained_leg(seconds=1.0) if (side and leg_on("sustained") and args.mode == "both") else None
    comm = (comm_leg_points(dense_flags) if by_points else comm_leg(dense_flags)) if shard is not None else None
    moved = moved_leg() if (side and leg_on("moved") and args.mode == "both") else None

    if rank == 0 and args.dump:
        np.savez(args.dump, scalars=out[0].cpu().numpy(), pg=out[1].cpu().numpy(), qg=out[2].cpu().numpy(), rewards=out[3].cpu().numpy())
    if rank == 0:
        # dominant kernel: pass 1, the one launch that evaluates every pair.  VALU-issue roofline (module docstring).
        mix = isa_mix()
        local_evals = cloud.n * n_virtual   # this rank's evaluations (its points x its virtual waypoints)
        evals_per_iter = 64 * mix["points_per_lane"]
        cyc_per_iter = (mix["packed_f32"] * ISSUE_CYCLES["packed_f32"] + mix["transcendental"] * ISSUE_CYCLES["transcendental"] +
                        mix["other_valu"] * ISSUE_CYCLES["other_valu"])
        p1_ms = kern[PASS1][0] / kern[PASS1][1]
        issue_cycles = cyc_per_iter * local_evals / evals_per_iter
        achieved = issue_cycles / (p1_ms * 1e-3) / 1e9           # G issue-cycles/s
        if not dense_flags:
            # --mode culled: pass 1 evaluates only the pairs a waypoint can reach (2.5 % on this workload); the priced instruction
            # stream is the dense kernel's, so the roofline of this line is scaled by the evaluated share (tohip_traj_step_stats)
            st = ops.traj_step_stats(cloud, ws)
            share = st["evaluated_pairs"] / max(1, st["slots"] * st["virtual_waypoints"])
            issue_cycles *= share
            achieved *= share
        peak = N_SIMDS * CLOCK_GHZ
        pmc = pmc_figures(PASS1)
        traffic, valu_busy, valu_busy_res = pmc.get("traffic"), pmc.get("valu_busy"), pmc.get("valu_busy_resident")
        step_ms = 1e3 * dt / args.steps
        line = {
            "metric": "point-visibility evals/sec (fwd+bwd)", "value": value, "unit": "evals/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "ms_per_step_windows": {"median": win_dense[len(win_dense) // 2], "min": win_dense[0], "max": win_dense[-1], "windows": len(win_dense),
                                    "note": "further windows of K steps each after the timed one (rank 0's clock)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.points}-point cloud x {args.wps_per_gpu} waypoints per GPU "
                                   f"({w_total} total)" + (f" x {args.cameras} cameras" if args.cameras > 1 else "") +
                                   ", fwd + bwd (x,y,z,quaternion) gradients",
                       "n_points": args.points, "waypoints_total": w_total, "cameras": args.cameras,
                       "parallelism": (f"point-shard x{n_gpus}" if by_points else f"waypoint-shard x{n_gpus}") if n_gpus > 1 else "single GPU",
                       "launch": "HIP graph replay of the step's launches" if use_graph else
                                 ("tohip_traj_forward_backward: one host call, five launches" if (shard is None and args.fused_step == "on") else
                                  ("tohip_traj_pshard_pass1 -> [MAX all-reduce, 16 B / waypoint] -> tohip_traj_pshard_local -> [SUM all-reduce, 320 B / "
                                   "waypoint] -> tohip_traj_pshard_finish: six launches" if by_points else
                                   "tohip_traj_forward -> [all-reduce] -> tohip_traj_reward_backward [-> all-gather]: five launches")),
                       "mode": "dense: every (point, waypoint) pair evaluated, no data-dependent skipping; the 0.7 % of "
                               "(256-point slot, waypoint) pairs that can contribute are then revisited by the sparse kernel",
                       "loss_vis": float(out[0][1].item())},
            "roofline": {"bound": "valu", "kernel": PASS1, "achieved": achieved, "peak": peak, "unit": "G VALU-issue-cycles/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "isa_mix_stale": bool(mix.get("stale")),   # True: the kernels' sources changed after the mix was counted
                         "kernel_ms_per_step_sum": sum(v[0] for v in kern.values()) / args.steps,
                         "kernel_ms_per_launch": p1_ms, "kernel_share_of_step": p1_ms / step_ms,
                         "issue_cycles_per_launch": issue_cycles,
                         "isa_mix_per_wave_iteration": mix, "issue_cycles_per_instruction": ISSUE_CYCLES,
                         "peak_note": f"{N_SIMDS} SIMDs x {CLOCK_GHZ} GHz; frac is a lower bound of the SIMDs' VALU-busy fraction "
                                      "(the chip clocks 2.2-2.4 GHz under this load; packed f32 measures 4.0-4.4 cycles)",
                         "valu_busy_pmc": valu_busy,
                         "valu_busy_while_resident_pmc": valu_busy_res,
                         "pmc_file": pmc.get("file"),
                         "profile_age": profile_age(pmc, p1_ms),
                         "valu_busy_note": "SQ_ACTIVE_INST_VALU x 4 over the SIMD cycles of the whole dispatch (GRBM_GUI_ACTIVE), and over the "
                                           "cycles the shader engines hold the kernel's waves (SQ_BUSY_CYCLES / 32 SEs): the difference, "
                                           "~12 % of a 100 us dispatch, is the kernel boundary (launch, cache invalidate / write-back, drain)",
                         "in_kernel": None if span_ms is None else {
                             "span_ms": span_ms, "clock_ghz": clock_ghz,
                             "frac_of_peak": issue_cycles / (span_ms * 1e-3) / 1e9 / peak,
                             "frac_at_measured_clock": issue_cycles / (span_ms * 1e-3) / 1e9 / (N_SIMDS * clock_ghz),
                             "note": "first block start -> last block end (s_memrealtime stamps inside the kernel, a separate "
                                     "diagnostic pass): what the blocks themselves take; kernel_ms_per_launch also holds the "
                                     "launch / queueing share of a back-to-back dependent launch"},
                         "hbm_counter_frac": (traffic / (p1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "traffic_note": "HBM bytes per launch from pmc_file (separate --pmc passes of "
                                         "this command: 2 x FETCH_SIZE + WRITE_SIZE); a few per cent of the HBM peak: points live "
                                         "in registers, waypoints stream through SGPRs, the 8-byte (min, max) stores go out as "
                                         "32-byte sectors",
                         "algorithmic_bytes_frac": value / n_gpus * ALGO_BYTES_FWD_BWD / (HBM_PEAK_GBS * 1e9),
                         "algorithmic_bytes_note": "48 B/eval of a streaming implementation (SURVEY.md 8d) x evals/s / 8 TB/s: above 1 "
                                                   "because nothing streams — kept for comparison with the survey's model only",
                         "kernel_ms": {k: v[0] / args.steps for k, v in kern.items()},
                         "kernel_ms_note": "per step and kernel class; HIP events on the launch stream over a second pass of the "
                                           "same K steps (the events themselves cost ~0.06 ms per step, so the timed pass runs "
                                           "without them)"},
        }
        same = all(torch.equal(a, b) for a, b in zip(out, out_c))
        line["culled_exact"] = {
            "value": evals_per_step * args.steps / dt_c, "unit": "evals/s", "ms_per_step": 1e3 * dt_c / args.steps,
            "bitwise_identical_to_dense": bool(same),
            "ms_per_step_windows": {"median": win_culled[len(win_culled) // 2], "min": win_culled[0], "max": win_culled[-1]},
            "kernel_ms": {k: v[0] / args.steps for k, v in kern_c.items()},
            "note": "library default (what ModelTraj runs): pass 1 skips pairs that provably can neither be a waypoint's maximum "
                    "nor contribute, via a Morton-sorted cloud, per-256-point bounding spheres and a distance bound on p"}
        if n_gpus > 1 or forced:
            line["ranks_seen"], line["backend"] = dist.get_world_size(), dist.get_backend()
        if sustained is not None:
            line["sustained"] = sustained
        if comm is not None:
            line["comm"] = comm
        if moved is not None:
            line["after_optimisation"] = moved

        def emit(name, obj):
            """A side leg's result: into the record, and on stdout as a line of its own BEFORE the headline."""
            line[name] = obj
            print(json.dumps({"detail": name, name: _short(obj)}, separators=(",", ":")), flush=True)
        if side:
            if leg_on("dropin"):
                emit("dropin", dropin_leg(device))
            if leg_on("configs"):
                emit("configs", configs_leg(device))
            if leg_on("message"):
                emit("message", message_leg(device))
            if leg_on("density"):
                emit("density_sweep", density_leg(device))
            if leg_on("aux"):
                emit("aux", aux_leg(device))
            if leg_on("occlusion"):
                emit("occlusion", occlusion_leg(device))
            if leg_on("hpr"):
                emit("hpr", hpr_leg(pts, device))
            if leg_on("cpu"):
                emit("cpu_baseline", cpu_baseline(args.points, w_total, args.cpu_wps))
            line["reference_cpu_container"] = {
                "value": 1.3e7, "unit": "evals/s", "cores": 8,
                "note": "the reference ITSELF (torch CPU, fwd+bwd, 1 M x 16) timed in the build container — it cannot travel to the "
                        "GPU box; profiles/r01_reference_cpu_timing.txt"}
        for name in ("sustained", "comm", "after_optimisation"):
            if name in line:
                print(json.dumps({"detail": name, name: _short(line[name])}, separators=(",", ":")), flush=True)
        print(json.dumps({"detail": "roofline_full", "roofline": _short(line["roofline"]), "culled_exact": _short(line["culled_exact"])},
                         separators=(",", ":")), flush=True)
        line["bench_wall_s"] = time.perf_counter() - t_bench0
        if args.details_file and args.details_file not in ("-", "none"):
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.details_file)), exist_ok=True)
                with open(args.details_file, "w") as f:
                    json.dump(line, f, indent=1)
                line["details_file"] = os.path.relpath(args.details_file, REPO)
            except OSError as e:   # a read-only tree: the earlier stdout lines still carry everything
                line["details_file"] = f"not written: {e}"[:80]
        print(compact_line(line), flush=True)
    if n_gpus > 1 or forced:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
