#!/usr/bin/env python3
"""Per (kernel, frames per launch) durations from a rocprofv3 --kernel-trace CSV: bench.py launches the
same kernel with 16 frames (the timed region) and with one frame (single_launch_us), which the --stats summary
averages together.  The tile and window kernels are templates <OutLens, InMode, Interp | QMode, CH, Frames, GeoRead[, SS]>: the FIFTH
argument says whether the instantiation has the frame loop (a wavefront walks `frames_per_wave` frames, grid y = groups of frames: its
launches are the batch launches, `batch` frames each — bench.py's 16); the sixth only says where the coordinates come from.  Every other instantiation renders grid y frames per launch (one for a single launch).
usage: kernel_trace_summary.py <kernel_trace.csv> [frames per launch of the frame-loop instantiations: 16]"""
import collections
import csv
import re
import sys


def frames_per_launch(name, grid_y, batch):
    m = re.search(r"reproject_(?:bicubic_win|tile)_kernel<([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(1).split(",")]
        if len(args) >= 6 and args[4] == "true":  # the frame loop
            return batch
    return grid_y


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    groups = collections.defaultdict(list)
    for r in rows:
        if "reproject" not in r["Kernel_Name"] and "corner_fill" not in r["Kernel_Name"]:
            continue
        grid_y = int(r["Grid_Size_Y"]) // max(1, int(r.get("Workgroup_Size_Y", 1) or 1))
        frames = frames_per_launch(r["Kernel_Name"], grid_y, batch)
        groups[(r["Kernel_Name"], frames)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'kernel':78s} {'frames/launch':>13s} {'launches':>8s} {'avg us':>10s} {'min us':>10s} {'us/frame':>9s}")
    for (name, frames), v in sorted(groups.items()):
        avg = sum(v) / len(v) / 1e3
        print(f"{name[:78]:78s} {frames:13d} {len(v):8d} {avg:10.1f} {min(v) / 1e3:10.1f} {avg / frames:9.1f}")


if __name__ == "__main__":
    main()
