#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under rocprofv3 --pmc FETCH_SIZE, then
--pmc WRITE_SIZE; see tools/collect_traffic.sh).  For every bench workload: one launch of the
calibration kernel (stand-alone post_process on a 4096^2 RGBA frame: reads and writes every byte
of the 256 MiB frame exactly once with 16-byte accesses, so its byte counts are known — it also
separates the workloads in the dispatch sequence), then REPS frames of the workload's dominant
kernel on device-resident frames (a frame of the cubemap workload is six launches), cycling over
more distinct frames than the 256 MiB Infinity Cache holds.  The order is written next to the
counters (argv[1]) for tools/traffic_summary.py."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

REPS = 7  # the first frame of each workload is dropped by the summary (table builds, cold caches)
order_path = sys.argv[1]
names = sys.argv[2:] or [n for n in bench.WORKLOADS]
pkg = importlib.import_module("image-lens-reproject_amd")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
cal = torch.empty((4096, 4096, 4), dtype=torch.float32, device=dev)
pkg.synth_fill(cal, 4096, 4096, 4, 0x5EED0000)
order = []
for name in names:
    wl = bench.WORKLOADS[name]
    size = wl["size"]
    n_res = 2 if wl.get("faces") else 3
    srcs, dsts = bench.resident_frames(torch, pkg, wl, size, n_res, dev, 0x5EED2000)
    out_size = dsts[0][0].shape[0]
    c = wl["channels"]
    lin, lout = bench.make_lens(pkg, wl["in_lens"], size, size), bench.make_lens(pkg, wl["out_lens"], out_size, out_size)
    torch.cuda.synchronize()
    pkg.post_process(pkg.Image(pkg.LensInfo.equirectangular(), 4096, 4096, 4, cal), 1.0009765625, 4.0)  # separator + calibration
    torch.cuda.synchronize()
    for i in range(REPS):
        im_in = pkg.Image(lin, size, size, c, srcs[i % n_res])
        outs = [pkg.Image(lout, out_size, out_size, c, d) for d in dsts[i % n_res]]
        if wl.get("faces"):
            pkg.reproject_multi(im_in, outs, 1, wl["interp"], np.stack([bench.make_rot(pkg, f) for f in wl["faces"]]), post=wl.get("post"))
        else:
            pkg.reproject(im_in, outs[0], 1, wl["interp"], bench.make_rot(pkg, wl["rot"]), post=wl.get("post"))
        torch.cuda.synchronize()
    order.append({"workload": name, "frames": REPS, "launches_per_frame": len(wl.get("faces") or [None])})
    del srcs, dsts
    torch.cuda.empty_cache()
with open(order_path, "w") as f:
    json.dump(order, f)
print("traffic probe done")
