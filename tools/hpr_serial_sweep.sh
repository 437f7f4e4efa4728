#!/bin/bash
# Run ON THE GPU BOX: the sample phase by sequential insertion (k_sample_hull, r06) against the sample rounds it replaces, and the
# knob that sizes it (face ids per segment), on the 1 M-point build and the 128-view batched build.  -> gpurun_out/hpr_serial_sweep.txt
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/hpr_serial_sweep.txt
: > $out
for cfg in "TOHIP_HULL_SERIAL=0" "TOHIP_HULL_SERIAL_IDS=64" "TOHIP_HULL_SERIAL_IDS=128" "TOHIP_HULL_SERIAL_IDS=192" "TOHIP_HULL_SERIAL_IDS=256" "TOHIP_HULL_SERIAL_IDS=320" "TOHIP_HULL_SERIAL_IDS=512" "TOHIP_HULL_SERIAL_IDS=768" "TOHIP_HULL_SERIAL_IDS=1024"; do
  echo "== $cfg" >> $out
  env $cfg python3 tools/hpr_once.py 1000000 12 2>/dev/null | tail -1 >> $out || exit 1
  env $cfg python3 tools/hpr_batched_once.py 4 2>/dev/null | tail -1 >> $out || exit 1
done
cat $out
