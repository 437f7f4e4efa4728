#!/usr/bin/env python3
"""Why the reference's loop replayed from a HIP graph is not faster than issuing it (bench.py -> dropin): the captured step
(model(); backward(); torch.optim.Adam(capturable=True).step()) on the 1 M x 128 workload, replayed N times, for

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pg -o g -- python3 tools/prof_graph.py 100
    python tools/prof_graph.py --summarize gpurun_out/pg/g_kernel_trace.csv     -> profiles/r04_graph_replay.txt

The summary counts, per replay, the kernels, their summed durations and the span from the first kernel's start to the last one's
end: a graph replay cannot go faster than that span, whatever the host does."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarize(path):
    import collections
    import csv
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:70]))
    rows.sort()
    # a replay starts at k_traj_probe; keep the replays of the second half of the trace (the timed ones)
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_traj_probe")]
    starts = starts[len(starts) // 2:]
    spans, sums, counts = [], [], []
    names = collections.Counter()
    per = collections.defaultdict(float)
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b]
        spans.append((rows[b][0] - seg[0][0]) / 1e3)
        sums.append(sum(e - s for s, e, _ in seg) / 1e3)
        counts.append(len(seg))
        for s, e, n in seg:
            names[n] += 1
            per[n] += (e - s) / 1e3
    n = len(spans)
    print(f"{n} replays: {sum(counts) / n:.1f} kernels per replay, start-to-start {sum(spans) / n:.1f} us, kernel durations summed {sum(sums) / n:.1f} us")
    for k, c in names.most_common():
        print(f"  {c / n:5.1f} x {per[k] / c:7.2f} us  {k}")


if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    summarize(sys.argv[2])
    sys.exit(0)

import torch  # noqa: E402
from trajectory_optimization_amd import synth  # noqa: E402
from trajectory_optimization_amd.model import ModelTraj  # noqa: E402

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
pts = torch.from_numpy(synth.make_cloud(1_000_000, seed=0)).to(dev)
poses, quats = synth.make_path(128, optical=True)
m = ModelTraj(pts, torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(synth.K_INTRINS), synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev)
opt = torch.optim.Adam([{"params": [m.poses], "lr": 0.1}, {"params": [m.quats], "lr": 0.02}], capturable=True)
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(3):
        opt.zero_grad()
        m(vis_wps_dist=0.0).backward()
        opt.step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize(dev)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss = m(vis_wps_dist=0.0)
    loss.backward()
    opt.step()
for _ in range(replays):
    g.replay()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(replays):
    g.replay()
torch.cuda.synchronize(dev)
print(f"{1e3 * (time.perf_counter() - t0) / replays:.4f} ms per replay (wall)")
