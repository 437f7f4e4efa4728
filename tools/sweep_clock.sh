#!/bin/bash
cd $GRAFT_REPO_ROOT
for NB in 512 1024 1280 1536 2048; do
  echo -n "DENSE_BLOCKS=$NB: "
  TOHIP_DENSE_BLOCKS=$NB timeout -k 5 60 python tools/pass1_clock.py 2>/dev/null
done
