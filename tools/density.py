#!/usr/bin/env python3
"""bench.py's density sweep alone (GPU box)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for r in bench.density_leg(torch.device("cuda:0")):
    print(json.dumps(r))
