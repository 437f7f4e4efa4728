"""bench.py's output contract, checked without a GPU: the LAST stdout line is one compact, strictly parsable JSON object well below
the 8 KB tail the driver keeps (BENCH_r05.json: `parsed: null` — the r05 line was 22 KB); `--gpus N` started without a launcher
spawns its ranks as children, relays rank 0's headline as its own last line and exits with their status; the CPU-baseline protocol
(median of >= 5 repetitions, per-repetition times, cores = min(affinity, cgroup quota))."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402  (imports torch, never the GPU)


def _full_record():
    """The r05 run's full record (gpurun_out/r5_bench_final.json: every side leg on, 22 KB) — the worst case the line was cut from."""
    with open(os.path.join(REPO, "tests", "golden", "bench_full_record_r05.json")) as f:
        return json.load(f)


def test_headline_line_is_compact_and_strict_json():
    full = _full_record()
    assert len(json.dumps(full)) > 20_000
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) < 8000 and len(line.encode()) <= bench.HEADLINE_BUDGET
    d = json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "culled_exact", "sustained"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-5) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert d["config"]["workload"].startswith("1000000-point cloud x 128 waypoints")
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch", "valu_busy_pmc"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["culled_exact"]["bitwise_identical_to_dense"] is True
    # the side legs keep one or two figures each
    assert d["hpr"]["gpu_ms"] > 0 and d["aux_16M_points"]["pointcloud2_to_xyz"]["frac_of_hbm_peak"] > 0


def test_headline_line_sheds_optional_objects_before_it_outgrows_the_tail():
    full = _full_record()
    full["comm"] = {f"collective_{i}": {"ms_median": 0.0123456789, "bytes": 4_000_000, "note": "x" * 50} for i in range(300)}
    line = bench.compact_line(full)
    assert len(line.encode()) <= bench.HEADLINE_BUDGET
    d = json.loads(line)
    assert "comm" in d["dropped_from_headline"] and "roofline" in d and "cpu_baseline" in d and d["value"] > 0


def test_non_finite_numbers_never_reach_the_line():
    full = _full_record()
    full["roofline"]["traffic"] = float("nan")
    full["hpr"]["gpu_ms"] = float("inf")
    d = json.loads(bench.compact_line(full), parse_constant=lambda c: pytest.fail(c))
    assert d["roofline"]["traffic"] is None and d["hpr"]["gpu_ms"] is None


def test_profile_age_flags_a_stale_counter_file():
    assert bench.profile_age({"kernel_ms": 0.0930}, 0.0975)["stale"] is False
    old = bench.profile_age({"kernel_ms": 0.0930}, 0.0700)
    assert old["stale"] is True and old["checked"] and old["ratio"] == pytest.approx(0.0700 / 0.0930)
    assert bench.profile_age({}, 0.09) == {"checked": False}
    pmc = bench.pmc_figures("k_traj_pass1")
    assert pmc["file"].startswith("profiles/r") and pmc["traffic"] > 1e6 and 0.05 < pmc["kernel_ms"] < 0.2


_FAKE_RANK = textwrap.dedent("""
    import json, os, sys
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    assert "--gpus" in sys.argv and sys.argv[sys.argv.index("--gpus") + 1] == str(world)
    if rank == 0:
        print(json.dumps({"detail": "leg", "leg": [1, 2, 3]}), flush=True)
        print(json.dumps({"metric": "m", "value": 1.0, "n_gpus": world, "ranks_seen": world}), flush=True)
        print("a line after the headline", flush=True)
    sys.exit(int(os.environ.get("FAKE_FAIL_RANK", "-1")) == rank)
""")


def _spawn(tmp_path, fail_rank=None):
    script = tmp_path / "fake_rank.py"
    script.write_text(_FAKE_RANK)
    code = ("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.spawn_ranks(2, argv=['--gpus', '2', '--steps', '3'], script=%r))"
            % (REPO, str(script)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    if fail_rank is not None:
        env["FAKE_FAIL_RANK"] = str(fail_rank)
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)


def test_gpus_n_spawns_its_own_ranks_and_relays_rank0(tmp_path):
    r = _spawn(tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l]
    last = json.loads(lines[-1])                       # the headline is the LAST line whatever was printed after it
    assert last == {"metric": "m", "value": 1.0, "n_gpus": 2, "ranks_seen": 2}
    assert json.loads(lines[0])["detail"] == "leg" and "a line after the headline" in lines[:-1]


def test_a_failing_rank_fails_the_parent(tmp_path):
    r = _spawn(tmp_path, fail_rank=1)
    assert r.returncode != 0


def test_cpu_baseline_protocol():
    from oracle import cpu_baseline as cb
    reps = [0.52, 0.50, 0.61, 0.51, 0.50, 0.53]
    b = cb.summarise(reps, 32_000_000, 16, 256, 256, 16.0, 1_000_000, 32, 0.9)
    assert b["value"] == pytest.approx(32_000_000 / 0.515) and b["reps"] == 6 and b["cores"] == 16 and b["kind"] == "port"
    assert b["rep_s_min"] == 0.50 and b["rep_s_max"] == 0.61 and b["spread_max_over_min"] == pytest.approx(1.22)
    assert b["nproc"] == 256 and b["affinity"] == 256 and b["cgroup_quota_cores"] == 16.0 and len(b["rep_s"]) == 6
    n, aff, nproc, quota = cb.usable_cores()
    assert 1 <= n <= aff <= nproc and (quota is None or n <= quota + 1e-6)
    env = cb.child_env()
    assert env["OMP_NUM_THREADS"] == str(n) and env["OMP_PROC_BIND"] in ("close", os.environ.get("OMP_PROC_BIND")) and "OMP_PLACES" in env


def test_cpu_baseline_child_runs_and_prints_one_json_line():
    from oracle import cpu_baseline as cb
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--points", "20000", "--waypoints", "8", "--sample-waypoints", "4",
                        "--min-seconds", "0.2", "--max-reps", "6"], cwd=REPO, env=cb.child_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads(r.stdout.strip().splitlines()[-1])
    assert b["value"] > 1e5 and 5 <= b["reps"] <= 6 and b["unit"] == "evals/s" and b["rep_s_min"] <= b["rep_s_max"]
    assert b["omp"]["OMP_PROC_BIND"] and b["omp"]["OMP_NUM_THREADS"] == str(b["cores"])
