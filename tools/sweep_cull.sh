#!/bin/bash
# run on the GPU box: culled pass-1 block-row width sweep
cd $GRAFT_REPO_ROOT
for VT in 8 16 32 64; do
  echo -n "CULL_VT=$VT: "
  TOHIP_CULL_VTILE=$VT MODES=cull timeout -k 5 60 python tools/time_traj.py 2>/dev/null | sed 's/.*cull: //' | cut -c1-230
done
