#!/usr/bin/env python3
"""Instruction mix of the gfx950 kernels: hipcc -S --cuda-device-only, then count mnemonics per kernel
(and inside the hottest loop).  Usage: tools/isa_stats.py [kernel-name-substring ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "trajectory_optimization_amd", "csrc", "trajopt_hip.hip")


def main():
    pats = sys.argv[1:] or ["k_traj_pass1ILi4ELb1", "k_traj_pass2ILi4ELb1ELb0", "k_traj_bwdILi4ELb1"]
    out = os.path.join(tempfile.gettempdir(), "trajopt_isa.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
                           "-S", "--cuda-device-only", SRC, "-o", out])
    text = open(out).read().split("\n")
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\w+):", text[i])
        if m and any(p in m.group(1) for p in pats):
            name = m.group(1)
            j = i + 1
            body = []
            while j < len(text) and "s_endpgm" not in text[j]:
                body.append(text[j])
                j += 1
            ins = []
            for l in body:
                l = l.strip()
                if not l or l.startswith((".", ";", "/")) or l.endswith(":"):
                    continue
                ins.append(l.split()[0])
            c = collections.Counter(ins)
            trans = sum(v for k, v in c.items() if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", k))
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            print(f"{name[:60]}: {len(ins)} instrs, VALU {valu} (transcendental {trans}), "
                  f"SALU/SMEM {sum(v for k, v in c.items() if k.startswith('s_'))}, "
                  f"VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_')))}, "
                  f"DS {sum(v for k, v in c.items() if k.startswith('ds_'))}")
            print("   ", dict(c.most_common(28)))
            i = j
        i += 1


if __name__ == "__main__":
    main()
