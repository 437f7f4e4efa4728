#!/usr/bin/env python3
"""Step time of the reference's own loop over the drop-in classes (bench.py's `dropin` object, alone).  Run on the GPU box."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.dropin_leg(torch.device("cuda:0")), indent=1))
