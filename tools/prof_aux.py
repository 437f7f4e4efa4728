#!/usr/bin/env python3
"""bench.py's aux leg alone (the HBM-bound kernels around the hot path), for rocprofv3:

    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_aux.py [points ...]     -> profiles/r04_aux_kernel_stats.csv
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ... / --pmc WRITE_SIZE ...              -> profiles/r04_aux_pmc.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sizes = tuple(int(a) for a in sys.argv[1:]) or (1_000_000, 16_000_000)
for size, rows in bench.aux_leg(torch.device("cuda:0"), sizes=sizes, reps=10).items():
    if size == "note":
        continue
    for name, r in rows.items():
        print(f"{size:>16s}  {name:62s} {r['us_per_call']:9.1f} us  {r['GBps']:8.1f} GB/s  {100 * r['frac_of_hbm_peak']:5.1f} % of 8 TB/s")
