#!/bin/bash
# Run ON THE GPU BOX: what the point kernel of the batched hull (k_reassign_only, TOHIP_HULL_SPLIT_LINK=1) is made of — counter passes
# of their own (no trace domains besides --kernel-trace), one small group per run.  -> gpurun_out/pmc_reassign/*.csv
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_reassign
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export TOHIP_HULL_SPLIT_LINK=1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/g$i" -o pmc -- python3 "$root/tools/hpr_batched_once.py" 1 > "$out/g$i.log" 2>&1 || echo "group $i failed: $grp"
done
find "$out" -name "*counter_collection.csv" | head
