#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under rocprofv3 --pmc FETCH_SIZE, then
--pmc WRITE_SIZE; see tools/collect_traffic.sh).  The calibration kernel — stand-alone post_process on a 4096^2 RGBA
frame: reads and writes every byte of the 256 MiB frame exactly once with 16-byte accesses, so its byte counts are known —
also SEPARATES the groups of dispatches: for every bench workload
    separator, WARM single-frame launches (table builds, the launch that fills the geometry cache and builds its lists,
               the per-face launches of a cubemap's first call), separator, REPS single-frame launches of the steady state
               (whatever kernels a frame takes: one window launch, a fill launch in front of it, six launches for six faces),
    separator, one 16-frame launch to warm up, separator, BATCH_LAUNCHES 16-frame launches (lrp_reproject_batch_device: what
               bench.py times), cycling over more distinct frames than the 256 MiB Infinity Cache holds.
The order of the groups is written next to the counters (argv[1]) for tools/traffic_summary.py, which sums ALL dispatches of a
measured group and divides by its frames."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

WARM, REPS = 3, 6  # single-frame launches per workload: warm-up group, measured group
BATCH, BATCH_LAUNCHES = 16, 2  # then 16-frame launches (what bench.py times), behind one of warm-up
order_path = sys.argv[1]
names = sys.argv[2:] or [n for n in bench.WORKLOADS]
pkg = importlib.import_module("image-lens-reproject_amd")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
cal = torch.empty((4096, 4096, 4), dtype=torch.float32, device=dev)
pkg.synth_fill(cal, 4096, 4096, 4, 0x5EED0000)


def separator():
    torch.cuda.synchronize()
    pkg.post_process(pkg.Image(pkg.LensInfo.equirectangular(), 4096, 4096, 4, cal), 1.0009765625, 4.0)  # separator + calibration
    torch.cuda.synchronize()


order = []
for name in names:
    wl = bench.WORKLOADS[name]
    size = wl["size"]
    faces = wl.get("faces")
    n_res = 2 if faces else BATCH
    srcs, dsts = bench.resident_frames(torch, pkg, wl, size, n_res, dev, 0x5EED2000)
    out_size = dsts[0][0].shape[0]
    c = wl["channels"]
    lin, lout = bench.make_lens(pkg, wl["in_lens"], size, size), bench.make_lens(pkg, wl["out_lens"], out_size, out_size)
    ins = [pkg.Image(lin, size, size, c, s) for s in srcs]
    outs = [[pkg.Image(lout, out_size, out_size, c, d) for d in ds] for ds in dsts]
    def single(i):
        if faces:
            pkg.reproject_multi(ins[i % n_res], outs[i % n_res], wl.get("ns", 1), wl["interp"], np.stack([bench.make_rot(pkg, f) for f in faces]), post=wl.get("post"))
        else:
            pkg.reproject(ins[i % n_res], outs[i % n_res][0], wl.get("ns", 1), wl["interp"], bench.make_rot(pkg, wl["rot"]), post=wl.get("post"))
        torch.cuda.synchronize()

    separator()
    for i in range(WARM):
        single(i)
    order.append({"workload": None})
    separator()
    for i in range(REPS):
        single(WARM + i)
    order.append({"workload": name, "frames": REPS, "launches_per_frame": "as dispatched"})
    if not faces:
        def batched():
            pkg.reproject_batch(ins, [o[0] for o in outs], wl.get("ns", 1), wl["interp"], bench.make_rot(pkg, wl["rot"]), post=wl.get("post"))
            torch.cuda.synchronize()

        separator()
        batched()
        order.append({"workload": None})
        separator()
        for i in range(BATCH_LAUNCHES):
            batched()
        order.append({"workload": name + "@batch16", "frames": BATCH_LAUNCHES, "launches_per_frame": "as dispatched"})
    del srcs, dsts, ins, outs
    torch.cuda.empty_cache()
with open(order_path, "w") as f:
    json.dump(order, f)
print("traffic probe done")
