#!/bin/bash
# Run ON THE GPU BOX: a diagnostic build of the library (-DTOHIP_SH_STAMPS) that prints where k_sample_hull's time goes per wave.
root=${GRAFT_REPO_ROOT:-.}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared -DTOHIP_SH_STAMPS $root/trajectory_optimization_amd/csrc/trajopt_hip.hip -o /tmp/libtrajopt_stamps.so || exit 1
TOHIP_LIB=/tmp/libtrajopt_stamps.so TOHIP_HULL_SERIAL_IDS=${1:-1024} python3 $root/tools/hpr_once.py 1000000 1 2>&1 | grep -v amdgpu.ids
