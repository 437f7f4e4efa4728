#!/bin/bash
# HPR build experiments, one environment variable at a time:
#   TOHIP_HULL_HBITS       bits of the apex height in a candidate's rank (0 = hashed order only)
#   TOHIP_HULL_JOIN_FACES  faces per segment at which all points join the sample's hull;  TOHIP_HULL_SUB  sample stride
VAR=${VAR:-TOHIP_HULL_HBITS}
for v in ${VALUES:-0 4 6 9 12 16}; do
  echo "== $VAR=$v"
  env $VAR=$v TOHIP_HULL_TRACE=1 timeout -k 10 120 python tools/hpr_outliers.py 1000000 12 2>&1 | grep "n=\|hull: [0-9]* rounds" | awk '/rounds;/{r+=$2; n++} /n=/{print} END{print "mean rounds", r/n}'
  env $VAR=$v timeout -k 10 120 python tools/hpr_batched_once.py 5 2>/dev/null || exit 1
done
