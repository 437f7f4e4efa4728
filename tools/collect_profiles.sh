#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 summaries of the bench command, written under gpurun_out/prof_<tag>/.
#   kernel-trace + stats for the dense and the culled mode, then counter passes of the dense mode in their own runs
#   (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ_* in a third), as MI355X_MICROARCH.md prescribes; the whole optimisation
#   loop (tools/prof_opt.py) and the streaming kernels around the path at 16 M points (tools/prof_aux.py) the same way.
# Afterwards, here:  python tools/summarize_profiles.py gpurun_out/prof_<tag> rNN   -> profiles/rNN_*
set -euo pipefail
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
common="--steps 20 --warmup 5 --cpu-wps 0 --details off --details-file none"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/dense" -o dense -- python3 "$root/bench.py" $common --mode dense > "$out/dense.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/culled" -o culled -- python3 "$root/bench.py" $common --mode culled > "$out/culled.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o pmc -- python3 "$root/bench.py" $common --mode dense > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o pmc -- python3 "$root/bench.py" $common --mode dense > "$out/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_sq" -o pmc -- python3 "$root/bench.py" $common --mode dense > "$out/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/hpr" -o hpr -- python3 "$root/tools/hpr_batched_once.py" 3 > "$out/hpr.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/hpr1m" -o hpr1m -- python3 "$root/tools/hpr_once.py" 1000000 3 > "$out/hpr1m.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/refresh" -o refresh -- python3 "$root/tools/prof_refresh.py" 4 hpr > "$out/refresh.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/opt" -o opt -- python3 "$root/tools/prof_opt.py" --steps 120 > "$out/opt.log" 2>&1
for sc in multi8 w1024 cam5 c2; do   # the large-W regime on one GPU (tools/prof_multi.py), culled = the library default
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/multi_$sc" -o multi_$sc -- python3 "$root/tools/prof_multi.py" --scenario $sc --mode culled --steps 40 --warmup 10 --no-events > "$out/multi_$sc.log" 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/message" -o message -- python3 "$root/tools/prof_message.py" 3 > "$out/message.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/aux" -o aux -- python3 "$root/tools/prof_aux.py" 16000000 > "$out/aux.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/aux_fetch" -o pmc -- python3 "$root/tools/prof_aux.py" 16000000 > "$out/aux_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/aux_write" -o pmc -- python3 "$root/tools/prof_aux.py" 16000000 > "$out/aux_write.log" 2>&1
find "$out" -name "*.csv" | head -60
