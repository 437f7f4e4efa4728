#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> duration statistics per (kernel, grid size): one kernel launched at several problem sizes
in the same run (tools/prof_aux.py) is several rows.   python tools/trace_by_grid.py <*_kernel_trace.csv> [name-substring ...]"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) > 2 and not any(w in name for w in sys.argv[2:]):
        continue
    rows[(name, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':48s} {'threads':>10s} {'block':>6s} {'calls':>6s} {'mean us':>9s} {'min us':>8s} {'median us':>9s} {'max us':>8s}")
for (name, grid, wg), d in sorted(rows.items()):
    d.sort()
    print(f"{name[:48]:48s} {grid:10d} {wg:6d} {len(d):6d} {sum(d) / len(d) / 1e3:9.2f} {d[0] / 1e3:8.2f} {d[len(d) // 2] / 1e3:9.2f} {d[-1] / 1e3:8.2f}")
