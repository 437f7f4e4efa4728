#!/bin/bash
# run on the GPU box: pass-1 tile / points-per-lane sweep
cd $GRAFT_REPO_ROOT
for P in 4 8; do
  for VT in 2 4 8 16 32 64 128; do
    echo -n "P=$P VT=$VT: "
    TOHIP_FORCE_P=$P TOHIP_FORCE_VTILE=$VT MODES=dense timeout -k 5 60 python tools/time_traj.py 2>/dev/null | sed 's/.*dense: //' | cut -c1-150
  done
done
