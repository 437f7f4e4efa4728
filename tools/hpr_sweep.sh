#!/bin/bash
# HPR build experiments: faces per segment at which all points join the sample's hull (TOHIP_HULL_JOIN_FACES), sample stride (TOHIP_HULL_SUB)
for j in ${JOINS:-768 1536}; do for s in ${SUBS:-96 128 192 256 512}; do
  echo "== join $j sub $s"
  TOHIP_HULL_JOIN_FACES=$j TOHIP_HULL_SUB=$s timeout -k 10 120 python tools/hpr_outliers.py 1000000 16 2>/dev/null || exit 1
  TOHIP_HULL_JOIN_FACES=$j TOHIP_HULL_SUB=$s timeout -k 10 120 python tools/hpr_batched_once.py 6 2>/dev/null || exit 1
done; done
