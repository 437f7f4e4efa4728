#!/bin/bash
# HPR build experiments: rounds per readback x batches per compaction (TOHIP_HULL_BATCH, TOHIP_HULL_COMPACT)
for b in 4 8; do for c in 1 2 4; do
  echo "== batch $b compact $c"
  TOHIP_HULL_BATCH=$b TOHIP_HULL_COMPACT=$c timeout -k 10 120 python tools/hpr_once.py 1000000 10 || exit 1
  TOHIP_HULL_BATCH=$b TOHIP_HULL_COMPACT=$c timeout -k 10 120 python tools/time_hpr_batched.py | tail -2 || exit 1
done; done
