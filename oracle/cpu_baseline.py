"""bench.py's `cpu_baseline` leg, as a process of its own:  python -m oracle.cpu_baseline --points N --waypoints W --sample-waypoints S

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): the CPU oracle (oracle/vis_oracle.c, f32, OpenMP) timed on a
bounded sample of bench.py's workload, forward + backward over S of the W waypoints.  A process of its own because the OpenMP runtime
reads OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES when it is loaded: bench.py starts this with them in the environment, and no torch
thread pool shares the cores.  Never touches the GPU (no torch import).

Protocol: one untimed warm-up repetition (pages, thread pool), then >= 5 timed repetitions (more until --min-seconds
of timed work, at most --max-reps); value = sample evaluations / MEDIAN repetition time; every repetition's time, min and max on the line.
Threads = min(cores this process may run on, the cgroup's CPU quota): a GPU box is a slice of a larger host and a pool sized by the
host's core count is throttled by the quota — that was the 2.7x swing of the earlier rounds' figure.
Prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time


def usable_cores():
    """(threads to use, affinity count, nproc, cgroup quota in cores or None)."""
    nproc = os.cpu_count() or 1
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                               # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    n = aff if quota is None else max(1, min(aff, int(math.floor(quota + 1e-9))))
    return n, aff, nproc, quota


def idlest_cores(n, sample_s=0.25):
    """The n allowed cores that were least busy over the last `sample_s` seconds (/proc/stat), as a sorted list — a GPU box is a slice of
    a larger host whose other tenants tend to sit on the first cores; None where /proc/stat cannot say."""
    def snap():
        out = {}
        with open("/proc/stat") as f:
            for line in f:
                if line.startswith("cpu") and line[3].isdigit():
                    p = line.split()
                    v = [int(x) for x in p[1:9]]
                    out[int(p[0][3:])] = (sum(v), v[3] + v[4])   # (total, idle + iowait)
        return out
    try:
        allowed = sorted(os.sched_getaffinity(0))
        a = snap()
        time.sleep(sample_s)
        b = snap()
        busy = {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in allowed if c in a and c in b}
        if len(busy) < n:
            return None
        # a contiguous run of allowed cores (neighbours share caches): the run of n with the lowest total load
        best, best_load = None, None
        for i in range(0, len(allowed) - n + 1):
            run = allowed[i:i + n]
            load = sum(busy.get(c, 1.0) for c in run)
            if best_load is None or load < best_load - 1e-9:
                best, best_load = run, load
        return best
    except (OSError, ValueError, KeyError, IndexError):
        return None


def child_env(threads=None):
    """The environment bench.py starts this module with (set BEFORE the OpenMP runtime loads): one thread per usable core, bound
    (OMP_PROC_BIND=close) to explicit places — the least busy contiguous run of allowed cores, or OMP_PLACES=cores where that
    cannot be measured."""
    n, _, _, _ = usable_cores()
    n = threads or n
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(n)
    env.setdefault("OMP_PROC_BIND", "close")
    if "OMP_PLACES" not in env:
        cores = idlest_cores(n)
        env["OMP_PLACES"] = ",".join("{%d}" % c for c in cores) if cores else "cores"
    env.setdefault("OMP_DYNAMIC", "false")
    return env


def summarise(rep_s, evals_per_rep, threads, aff, nproc, quota, n_points, n_sample, warmup_s):
    """The cpu_baseline object from the repetition times (kept apart from the timing so that a CPU test can check the protocol)."""
    t = sorted(rep_s)
    med = t[len(t) // 2] if len(t) % 2 else 0.5 * (t[len(t) // 2 - 1] + t[len(t) // 2])
    return {"value": evals_per_rep / med, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "protocol": "median of timed repetitions after one warm-up",
            "reps": len(t), "rep_s": [round(x, 4) for x in rep_s], "rep_s_min": t[0], "rep_s_max": t[-1], "spread_max_over_min": t[-1] / t[0],
            "warmup_s": round(warmup_s, 4), "nproc": int(nproc), "affinity": int(aff), "cgroup_quota_cores": quota,
            "omp": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")},
            "sample": f"{n_points} points x {n_sample} waypoints fwd+bwd per repetition, oracle f32 (C + OpenMP, {threads} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--waypoints", type=int, default=128)
    ap.add_argument("--sample-waypoints", type=int, default=32)
    ap.add_argument("--min-reps", type=int, default=5)
    ap.add_argument("--max-reps", type=int, default=15)
    ap.add_argument("--min-seconds", type=float, default=8.0)
    a = ap.parse_args()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import numpy as np
    from trajectory_optimization_amd import synth
    from oracle import oracle
    threads, aff, nproc, quota = usable_cores()
    threads = int(os.environ.get("OMP_NUM_THREADS", threads))
    pts = synth.make_cloud(a.points, seed=0)
    poses, quats = synth.make_path(a.waypoints, optical=True)
    sel = np.linspace(0, len(poses) - 1, a.sample_waypoints).astype(int)
    p, q = poses[sel], quats[sel]
    K, iw, ih = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT

    def rep():
        t0 = time.perf_counter()
        f = oracle.traj_forward(pts, p, q, K, iw, ih)
        oracle.traj_backward(pts, p, q, K, iw, ih, f)
        return time.perf_counter() - t0
    warm = rep()
    times = []
    while len(times) < a.min_reps or (sum(times) < a.min_seconds and len(times) < a.max_reps):
        times.append(rep())
    print(json.dumps(summarise(times, a.points * a.sample_waypoints, threads, aff, nproc, quota, a.points, a.sample_waypoints, warm)), flush=True)


if __name__ == "__main__":
    main()
