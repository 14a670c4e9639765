#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under rocprofv3 --pmc FETCH_SIZE, then
--pmc WRITE_SIZE; see tools/collect_traffic.sh).  Launches, on device-resident 4096^2
RGBA frames: the calibration kernel (stand-alone post_process: reads and writes every
byte of a 256 MiB frame exactly once with 16-byte accesses, so its byte counts are
known) and each bench workload's dominant kernel, a few times each."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import torch  # noqa: E402

pkg = importlib.import_module("image-lens-reproject_amd")
dev = torch.device("cuda", 0)
size, c = 4096, 4
# more distinct frames than the 256 MiB Infinity Cache holds
srcs = [torch.empty((size, size, c), dtype=torch.float32, device=dev) for _ in range(3)]
dsts = [torch.empty((size, size, c), dtype=torch.float32, device=dev) for _ in range(3)]
for i, s in enumerate(srcs):
    pkg.synth_fill(s, size, size, c, 0x5EED0000 + i)
torch.cuda.synchronize()
for i in range(3):  # calibration: post_process_kernel, 268435456 B read + 268435456 B written per launch
    pkg.post_process(pkg.Image(pkg.LensInfo.equirectangular(), size, size, c, srcs[i].clone()), 2.0, 4.0)
torch.cuda.synchronize()
for name in sys.argv[1:] or sorted(bench.WORKLOADS):
    wl = bench.WORKLOADS[name]
    lin = bench.make_lens(pkg, wl["in_lens"], size, size)
    lout = bench.make_lens(pkg, wl["out_lens"], size, size)
    rot = bench.make_rot(pkg, wl["rot"])
    for i in range(6):
        pkg.reproject(pkg.Image(lin, size, size, c, srcs[i % 3]), pkg.Image(lout, size, size, c, dsts[i % 3]), 1,
                      wl["interp"], rot)
    torch.cuda.synchronize()
print("traffic probe done")
