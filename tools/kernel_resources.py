#!/usr/bin/env python3
"""Register / LDS / occupancy figures of the library's kernels as hipcc reports them (-Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [name-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "trajectory_optimization_amd", "csrc", "trajopt_hip.hip")


def main():
    out = os.path.join(tempfile.gettempdir(), "trajopt_res.o")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-c", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", SRC, "-o", out] + [a for a in sys.argv[1:] if a.startswith("-D")],
                       stderr=subprocess.PIPE, text=True)
    want = [a for a in sys.argv[1:] if not a.startswith("-D")]
    cur = None
    rows = {}
    for line in p.stderr.split("\n"):
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2)
            rows[cur] = {}
        elif cur:
            rows[cur][m.group(1).split(" [")[0]] = m.group(2)
    for name, r in rows.items():
        short = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
        if want and not any(w in short for w in want):
            continue
        g = lambda k: str(r.get(k))  # noqa: E731
        print(f"{short[:60]:60s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} sgpr {g('TotalSGPRs'):>4s} (spilled {g('SGPRs Spill'):>3s}) "
              f"vgpr spill {g('VGPRs Spill'):>3s} scratch {g('ScratchSize'):>4s} occ {g('Occupancy'):>2s} lds {g('LDS Size'):>6s}")

if __name__ == "__main__":
    main()
