#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/collect_profiles.sh into the committed summaries:
    python tools/summarize_profiles.py gpurun_out/prof_r02 r02
 -> profiles/rNN_bench_{dense,culled}_kernel_stats.csv, profiles/rNN_hpr_{batched,1m}_kernel_stats.csv (copies) and
    profiles/rNN_optimize_kernel_stats.csv (tools/prof_opt.py), profiles/rNN_aux_kernel_stats.csv + rNN_aux_pmc.json (tools/prof_aux.py) and
    profiles/rNN_bench_dense_pmc.json: per kernel, HBM bytes per launch = 2 x FETCH_SIZE (gfx950 counts a wide coalesced read
    at half its bytes, MI355X_MICROARCH.md) + WRITE_SIZE, both reported in KB, and the VALU busy fraction
    SQ_ACTIVE_INST_VALU / (32 x GRBM_GUI_ACTIVE)."""
import collections
import csv
import json
import os
import shutil
import sys


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc, dur


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def aux_pmc(src, path):
    """tools/prof_aux.py at 16 M points under FETCH_SIZE and WRITE_SIZE (two runs): HBM bytes per launch of every streaming kernel
    around the path, with the corrected read bytes (x2, see the module docstring) and the rate over the launch's duration."""
    out = {"command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace -- python3 tools/prof_aux.py 16000000  (two runs)",
           "units": "KB per launch as reported; hbm_bytes_per_launch_corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", "kernels": {}}
    ks = out["kernels"]
    for sub, n in (("aux_fetch", "FETCH_SIZE"), ("aux_write", "WRITE_SIZE")):
        acc, dur = per_kernel(os.path.join(src, sub, "pmc_counter_collection.csv"))
        for k, counters in acc.items():
            if "k_" not in k or "rocprim" in k or "at::" in k or n not in counters:
                continue
            e = ks.setdefault(short(k), {})
            e["launches"] = len(dur[k])
            e[f"duration_ns_mean_{sub}"] = sum(dur[k]) / len(dur[k])
            e[f"{n}_mean_per_launch"] = sum(counters[n]) / len(counters[n])
    for e in ks.values():
        if "FETCH_SIZE_mean_per_launch" in e and "WRITE_SIZE_mean_per_launch" in e:
            e["hbm_bytes_per_launch_corrected"] = (2.0 * e["FETCH_SIZE_mean_per_launch"] + e["WRITE_SIZE_mean_per_launch"]) * 1024.0
            e["hbm_GBps_over_traced_duration"] = e["hbm_bytes_per_launch_corrected"] / e["duration_ns_mean_aux_fetch"]
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(repo, "profiles")
    for mode in ("dense", "culled"):
        shutil.copy(os.path.join(src, mode, f"{mode}_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_{mode}_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "hpr", "hpr_kernel_stats.csv")):
        shutil.copy(os.path.join(src, "hpr", "hpr_kernel_stats.csv"), os.path.join(dst, f"{tag}_hpr_batched_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "hpr1m", "hpr1m_kernel_stats.csv")):
        shutil.copy(os.path.join(src, "hpr1m", "hpr1m_kernel_stats.csv"), os.path.join(dst, f"{tag}_hpr_1m_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "refresh", "refresh_kernel_stats.csv")):
        shutil.copy(os.path.join(src, "refresh", "refresh_kernel_stats.csv"), os.path.join(dst, f"{tag}_occlusion_refresh_kernel_stats.csv"))
    for sub, name in (("opt", "optimize"), ("aux", "aux")):
        if os.path.exists(os.path.join(src, sub, f"{sub}_kernel_stats.csv")):
            shutil.copy(os.path.join(src, sub, f"{sub}_kernel_stats.csv"), os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "message", "message_kernel_stats.csv")):   # tools/prof_message.py
        shutil.copy(os.path.join(src, "message", "message_kernel_stats.csv"), os.path.join(dst, f"{tag}_message_kernel_stats.csv"))
    for sc in ("multi8", "w1024", "cam5", "c2"):   # tools/prof_multi.py scenarios
        f = os.path.join(src, f"multi_{sc}", f"multi_{sc}_kernel_stats.csv")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{tag}_multi_{sc}_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "aux_fetch", "pmc_counter_collection.csv")):
        aux_pmc(src, os.path.join(dst, f"{tag}_aux_pmc.json"))
    out = {"command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_*> --kernel-trace -- python3 bench.py --steps 20 --warmup 5 "
                      "--cpu-wps 0 --details off --mode dense   (three separate runs, tools/collect_profiles.sh)",
           "units": "FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
           "kernels": {}}
    ks = out["kernels"]
    for sub, names in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]),
                       ("pmc_sq", ["SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])):
        acc, dur = per_kernel(os.path.join(src, sub, "pmc_counter_collection.csv"))
        for k, counters in acc.items():
            if "k_" not in k or "rocprim" in k or "at::" in k:
                continue
            e = ks.setdefault(short(k), {})
            e["launches"] = len(dur[k])
            e[f"duration_ns_mean_{sub}"] = sum(dur[k]) / len(dur[k])
            for n in names:
                if n in counters:
                    e[f"{n}_mean_per_launch"] = sum(counters[n]) / len(counters[n])
    for k, e in ks.items():
        if "FETCH_SIZE_mean_per_launch" in e and "WRITE_SIZE_mean_per_launch" in e:
            e["hbm_bytes_per_launch_corrected"] = (2.0 * e["FETCH_SIZE_mean_per_launch"] + e["WRITE_SIZE_mean_per_launch"]) * 1024.0
        if "SQ_ACTIVE_INST_VALU_mean_per_launch" in e and e.get("GRBM_GUI_ACTIVE_mean_per_launch"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles of VALU issue summed over the chip (x4 = SIMD cycles spent issuing vector
            # instructions: it equals the ISA-derived count, 4 cycles per packed/other, 8 per transcendental instruction);
            # GRBM_GUI_ACTIVE is the kernel's busy cycles summed over the 8 XCDs; 1024 SIMDs:
            #   busy = 4 * ACTIVE / (1024 * GUI / 8) = ACTIVE / (32 * GUI)
            e["valu_issue_cycles_per_launch"] = 4.0 * e["SQ_ACTIVE_INST_VALU_mean_per_launch"]
            e["valu_busy_fraction"] = e["SQ_ACTIVE_INST_VALU_mean_per_launch"] / (32.0 * e["GRBM_GUI_ACTIVE_mean_per_launch"])
            e["clock_ghz_from_grbm"] = e["GRBM_GUI_ACTIVE_mean_per_launch"] / 8.0 / e["duration_ns_mean_pmc_sq"]
            if e.get("SQ_BUSY_CYCLES_mean_per_launch"):
                # SQ_BUSY_CYCLES: cycles a shader engine's sequencer holds waves, summed over the 32 SEs (8 XCDs x 4).  VALU issue
                # over THOSE cycles = how busy the SIMDs are while the kernel's waves are resident; the rest of GRBM_GUI_ACTIVE
                # is the dispatch's boundary (launch, cache invalidate / write-back, drain)
                e["sq_busy_share_of_dispatch"] = e["SQ_BUSY_CYCLES_mean_per_launch"] / 32.0 / (e["GRBM_GUI_ACTIVE_mean_per_launch"] / 8.0)
                e["valu_busy_while_resident"] = 4.0 * e["SQ_ACTIVE_INST_VALU_mean_per_launch"] / 1024.0 / (e["SQ_BUSY_CYCLES_mean_per_launch"] / 32.0)
    with open(os.path.join(dst, f"{tag}_bench_dense_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, e in ks.items():
        print(k[:40], {kk: round(v, 3) if isinstance(v, float) else v for kk, v in e.items()
                       if kk in ("launches", "hbm_bytes_per_launch_corrected", "valu_busy_fraction")})


if __name__ == "__main__":
    main()
