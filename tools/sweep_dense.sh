#!/bin/bash
# run on the GPU box: persistent-grid size sweep for the dense pass 1
cd $GRAFT_REPO_ROOT
for NB in 1024 1280 1536 1792 2048; do
  echo "DENSE_BLOCKS=$NB: "
  TOHIP_DENSE_BLOCKS=$NB MODES=dense timeout -k 5 60 python tools/time_traj.py 2>/dev/null | sed 's/.*dense: //' | cut -c1-110
  TOHIP_DENSE_BLOCKS=$NB timeout -k 5 60 python tools/pass1_clock.py 2>/dev/null | sed -n 2,3p
done
