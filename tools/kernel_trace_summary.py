#!/usr/bin/env python3
"""Per (kernel, frames per launch) durations from a rocprofv3 --kernel-trace CSV: bench.py launches the
same kernel with 16 frames (the timed region, blockIdx.y = frame) and with one frame (single_launch_us),
which the --stats summary averages together.  The frame-loop instantiations (last template argument `true`) render a
whole batch with grid y = groups of frames: their launches are bench.py's 16-frame launches.
usage: kernel_trace_summary.py <kernel_trace.csv> [frames per launch of the frame-loop instantiations: 16]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
groups = collections.defaultdict(list)
for r in rows:
    if "reproject" not in r["Kernel_Name"]:
        continue
    frames = int(r["Grid_Size_Y"])
    if ", true>" in r["Kernel_Name"]:
        frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    groups[(r["Kernel_Name"], frames)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':78s} {'frames/launch':>13s} {'launches':>8s} {'avg us':>10s} {'min us':>10s} {'us/frame':>9s}")
for (name, frames), v in sorted(groups.items()):
    avg = sum(v) / len(v) / 1e3
    print(f"{name[:78]:78s} {frames:13d} {len(v):8d} {avg:10.1f} {min(v) / 1e3:10.1f} {avg / frames:9.1f}")
