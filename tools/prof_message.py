#!/usr/bin/env python3
"""The per-message path alone (bench.py's message_leg) for `rocprofv3 --kernel-trace --stats` -> profiles/r05_message_kernel_stats.csv:
    rocprofv3 --kernel-trace --stats -d gpurun_out/pmsg -o msg -- python3 tools/prof_message.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.message_leg(torch.device("cuda:0"), reps=int(sys.argv[1]) if len(sys.argv) > 1 else 3)))
