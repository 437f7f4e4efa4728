#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the culled step, fused and split.  Output: gpurun_out/prof_quick/*.csv
set -euo pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_quick
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
common="--steps 20 --warmup 5 --cpu-wps 0 --dropin off"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/fused" -o fused -- python3 "$root/bench.py" $common --mode ${1:-culled} > "$out/fused.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/split" -o split -- python3 "$root/bench.py" $common --mode ${1:-culled} --fused-step off > "$out/split.log" 2>&1
for m in fused split; do echo "== $m"; python3 - "$out/$m/${m}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "k_traj" in r["Name"] or "k_adam" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}')
PY
done
