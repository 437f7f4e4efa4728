#!/bin/bash
# run on the GPU box: persistent-grid size sweep for the dense pass 1
cd $GRAFT_REPO_ROOT
for NB in 1024 1536 1792 2048 2560 3072 4096; do
  echo -n "DENSE_BLOCKS=$NB: "
  TOHIP_DENSE_BLOCKS=$NB MODES=dense timeout -k 5 60 python tools/time_traj.py 2>/dev/null | sed 's/.*dense: //' | cut -c1-110
done
