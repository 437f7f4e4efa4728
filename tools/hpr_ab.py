"""A/B of HPR build settings on one box: python tools/hpr_ab.py "ENV=V ..." "ENV=V ..." [...]  (use "-" for the defaults).
Every setting runs tools/hpr_once.py (1 M points) and tools/hpr_batched_once.py (128 views) in a process of its own, ROUNDS times
in alternation (a box drifts by a few percent over a minute); the table holds each setting's best and median."""
import os, re, subprocess, sys, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUNDS = int(os.environ.get("AB_ROUNDS", "3"))
cfgs = sys.argv[1:] or ["-"]
res = {c: ([], []) for c in cfgs}
for r in range(ROUNDS):
    for c in cfgs:
        env = dict(os.environ)
        if c != "-":
            env.update(dict(kv.split("=", 1) for kv in c.split()))
        for k, (script, args) in enumerate(((("hpr_once.py"), ["1000000", "16"]), (("hpr_batched_once.py"), ["4"]))):
            if os.environ.get("AB_ONLY") and int(os.environ["AB_ONLY"]) != k:
                continue
            out = subprocess.run([sys.executable, os.path.join(root, "tools", script)] + args, env=env, capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(c, script, "FAILED", out.stderr[-500:]); sys.exit(1)
            line = out.stdout.strip().splitlines()[-1]
            m = re.search(r"ms=([0-9.]+)", line) or re.search(r"([0-9.]+) ms", line)
            res[c][k].append(float(m.group(1)))
for c in cfgs:
    s, b = res[c]
    f = lambda v: f"best {min(v):6.2f} median {statistics.median(v):6.2f}" if v else "-"
    print(f"{c:50s} single: {f(s)}   batched: {f(b)}", flush=True)
