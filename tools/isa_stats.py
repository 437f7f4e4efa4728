#!/usr/bin/env python3
"""Instruction mix of the gfx950 kernels: hipcc -S --cuda-device-only, then count mnemonics per kernel and inside its
innermost loop.

  tools/isa_stats.py [kernel-name-substring ...]          print the mixes
  tools/isa_stats.py --json profiles/r02_pass1_isa_mix.json
        write the VALU mix of ONE (wave, waypoint) iteration of k_traj_pass1's dense inner loop — what bench.py prices for the
        VALU-issue roofline (packed f32 / transcendental / other VALU; SALU, s_nop and memory instructions issue on other ports)
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "trajectory_optimization_amd", "csrc", "trajopt_hip.hip")
DENSE_PASS1 = "k_traj_pass1_denseILb0E"   # <OCC = false>
TRANS = re.compile(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_")


def source_hash():
    """sha256 of the sources the dense loop is compiled from: bench.py refuses a mix that was counted on other code."""
    import hashlib
    h = hashlib.sha256()
    for f in ("common.hpp", "traj_kernels.hip"):
        with open(os.path.join(REPO, "trajectory_optimization_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def disassemble():
    out = os.path.join(tempfile.gettempdir(), "trajopt_isa.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
                           "-S", "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(text):
    """name -> (all instructions, instructions of blocks inside a loop, {loop header: its instructions})"""
    res = {}
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\w+):", text[i])
        if not m:
            i += 1
            continue
        name, j, ins, loop, in_loop, by_header, hdr = m.group(1), i + 1, [], [], False, collections.defaultdict(list), None
        while j < len(text) and "s_endpgm" not in text[j]:
            l = text[j].strip()
            j += 1
            mb = re.match(r"^(\.LBB(\w+):|; %bb\.\d+:)", l)
            if mb:
                in_loop = "Loop" in l
                mh = re.search(r"Header=BB(\w+)", l)
                hdr = mh.group(1) if mh else (mb.group(2) if (in_loop and mb.group(2)) else None)
                continue
            if not l or l.startswith((".", ";", "/")) or l.endswith(":"):
                continue
            op = l.split()[0]
            ins.append(op)
            if in_loop:
                loop.append(op)
                by_header[hdr].append(op)
        res[name] = (ins, loop, by_header)
        i = j
    return res


def hot_loop_ops(text, kernel_prefix, marker="s_load_dwordx16"):
    """Mnemonics of the innermost loop of `kernel_prefix` that contains `marker`, WITHOUT its cold regions: a region skipped by an
    `s_cbranch_execz` that holds a global atomic and no store (pass 1 folds a wave's extrema into the waypoint's running ones and
    lists a candidate slot only when the probe's bounds are beaten: a few dozen times per waypoint and launch)."""
    i = next(k for k, l in enumerate(text) if l.startswith(kernel_prefix))
    j = i
    while "s_endpgm" not in text[j]:
        j += 1
    body = text[i:j]
    k = next(n for n, l in enumerate(body) if marker in l)
    hdr = None
    for n in range(k, -1, -1):
        m = re.search(r"in Loop: Header=BB(\w+)", body[n])
        if m:
            hdr = m.group(1)
            break
    lines = [l.strip() for l in body if l.strip()]
    # the loop's blocks, in layout order
    sel, keep = [], False
    for l in lines:
        if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", l):
            keep = f"Header=BB{hdr} " in l + " " or l.startswith(f".LBB{hdr}:")
            sel.append(("label", l.split(":")[0]))
            continue
        if keep and not l.startswith((".", ";", "/")) and not l.endswith(":"):
            sel.append(("op", l))
    cold = [False] * len(sel)
    for a, (kind, l) in enumerate(sel):
        if kind == "op" and l.startswith("s_cbranch_execz"):
            tgt = l.split()[1]
            b = next((n for n in range(a + 1, len(sel)) if sel[n] == ("label", tgt)), None)
            if b is None:
                # the skip target lies earlier in the layout (the loop's latch): the region runs to its own jump there
                b = next((n + 1 for n in range(a + 1, len(sel)) if sel[n] == ("op", "s_branch " + tgt)), None)
            if b is None:
                continue
            region = [x for kk, x in sel[a + 1:b] if kk == "op"]
            if any(x.startswith("global_atomic") for x in region) and not any(x.startswith("global_store") for x in region):
                for n in range(a + 1, b):
                    cold[n] = True
        if kind == "op" and l.startswith("s_cbranch_scc"):
            # the slot's MINIMUM: computed only while the probe has not exhibited a zero (a waypoint-uniform branch around a
            # region of v_min instructions); with one exact zero in the cloud — every BASELINE workload — it is never entered
            tgt = l.split()[1]
            b = next((n for n in range(a + 1, len(sel)) if sel[n] == ("label", tgt)), None)
            if b is None:
                continue
            region = [x for kk, x in sel[a + 1:b] if kk == "op"]
            if any(x.startswith("v_min_i32_dpp") for x in region) and not any(x.startswith(("v_max_i32_dpp", "global_", "v_pk_")) for x in region):
                for n in range(a + 1, b):
                    cold[n] = True
    # blocks laid out after the loop's back edge that only cold code jumps to (the compiler moves a cold region's inner
    # branches out of line): cold too.  A block = a label's ops up to the next label; it is cold when the op before it does not
    # fall through into it (an unconditional s_branch, or cold) and every branch to its label is cold (a block without a label
    # of its own is entered by falling through only).
    changed = True
    while changed:
        changed = False
        for a, (kind, l) in enumerate(sel):
            if kind != "label":
                continue
            b = next((n for n in range(a + 1, len(sel)) if sel[n][0] == "label"), len(sel))
            if b == a + 1 or all(cold[a + 1:b]):
                continue
            prev = next((n for n in range(a - 1, -1, -1) if sel[n][0] == "op"), None)
            falls = prev is not None and not (cold[prev] or sel[prev][1].startswith("s_branch"))
            srcs = [n for n, (kk, x) in enumerate(sel) if kk == "op" and x.startswith(("s_cbranch", "s_branch")) and x.split()[1] == l]
            entered = bool(srcs) or (prev is not None and cold[prev] and not sel[prev][1].startswith("s_branch"))
            if not falls and entered and all(cold[n] for n in srcs):
                for n in range(a + 1, b):
                    cold[n] = True
                changed = True
    return [l.split()[0] for n, (kind, l) in enumerate(sel) if kind == "op" and not cold[n]], sum(1 for n, (kind, _) in enumerate(sel) if kind == "op" and cold[n])


def classes(ops):
    c = collections.Counter(ops)
    trans = sum(v for k, v in c.items() if TRANS.match(k))
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    return c, dict(packed_f32=pk, transcendental=trans, other_valu=valu - pk - trans,
                   salu_smem=sum(v for k, v in c.items() if k.startswith("s_")),
                   vmem=sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))),
                   lds=sum(v for k, v in c.items() if k.startswith("ds_")))


def main():
    args = sys.argv[1:]
    text = disassemble()
    ks = kernels(text)
    if args and args[0] == "--json":
        name = next(n for n in ks if DENSE_PASS1 in n)
        # the waypoint loop: the one that loads a waypoint record (s_load_dwordx16), without its cold regions
        body, n_cold = hot_loop_ops(text, name)
        c, cl = classes(body)
        out = dict(kernel=name, points_per_lane=8, evaluations_per_iteration=512,
                   packed_f32=cl["packed_f32"], transcendental=cl["transcendental"], other_valu=cl["other_valu"],
                   salu_smem=cl["salu_smem"], vmem=cl["vmem"], cold_instructions_excluded=n_cold, source_hash=source_hash(),
                   note="VALU instructions of one (wave, waypoint) iteration of the dense inner loop (hipcc -O3 --offload-arch=gfx950, "
                        "ROCm 7.2); a lane owns 8 points = 4 packed pairs; per evaluation: 25 FMA-class operations, 4 transcendentals; "
                        "the regions a wave enters only when its slot beats the probe's bounds (atomicMin / atomicMax of the running extrema: a few dozen "
                        "times per waypoint and launch) and the slot-minimum region (entered only for a waypoint whose sample held no exact zero: none "
                        "on the BASELINE workloads) are not counted; 25 FMA-class operations per evaluation since r03",
                   mnemonics=dict(c.most_common()))
        with open(args[1], "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps({k: out[k] for k in ("packed_f32", "transcendental", "other_valu", "salu_smem", "vmem")}))
        return
    pats = args or ["k_traj_pass1_dense", "k_traj_pass1_cull", "k_traj_sparse", "k_traj_finish", "k_traj_probe"]
    for name, (ins, loop, _) in ks.items():
        if not any(p in name for p in pats):
            continue
        for what, ops in (("kernel", ins), ("loops", loop)):
            c, cl = classes(ops)
            print(f"{name[:70]} [{what}]: {len(ops)} instrs {cl}")
            print("   ", dict(c.most_common(24)))


if __name__ == "__main__":
    main()
