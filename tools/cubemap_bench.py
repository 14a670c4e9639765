#!/usr/bin/env python3
"""BASELINE configs[4] — one 8192^2 RGB panorama into six 2048^2 rectilinear faces, bicubic — through lrp_reproject_multi_device,
with the faces merged into one launch (default) and as six launches.  usage: cubemap_bench.py [reps]"""
import importlib, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module("image-lens-reproject_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, m, c, n_src = 8192, 2048, 3, 4
dev = torch.device("cuda", 0)
srcs = []
for k in range(n_src):
    s = torch.empty((n, n, c), dtype=torch.float32, device=dev)
    pkg.synth_fill(s, n, n, c, 0x5EED0000 + k, -1)
    srcs.append(s)
dsts = [[torch.empty((m, m, c), dtype=torch.float32, device=dev) for _ in range(6)] for _ in range(n_src)]
faces = [(0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (180.0, 0.0, 0.0), (270.0, 0.0, 0.0), (0.0, 90.0, 0.0), (0.0, -90.0, 0.0)]
rots = np.stack([pkg.rotation_matrix(*[d * math.pi / 180.0 for d in f]) for f in faces])
lin, lout = pkg.LensInfo.equirectangular(), pkg.LensInfo.rectilinear(18.0, 36.0, m, m)
ins = [pkg.Image(lin, n, n, c, s) for s in srcs]
outs = [[pkg.Image(lout, m, m, c, d) for d in ds] for ds in dsts]
stream = torch.cuda.Stream()


def run(i):
    pkg.reproject_multi(ins[i % n_src], outs[i % n_src], 1, 2, rots, stream=stream)


for fork, strip in ((1, 0), (0, 0), (2, 0), (1, 1), (1, 2), (1, 4), (1, 0)):
    pkg.debug_set("multi_fork", fork)
    pkg.debug_set("geo_strip", strip)
    with torch.cuda.stream(stream):
        for i in range(8):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            run(i)
        e1.record(stream)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    once = (n * n + 6 * m * m) * c * 4
    print(f"multi_fork {fork} geo_strip {strip}: {us:7.1f} us per cubemap   frac_source_once {once / (us * 1e-6) / 8e12:.3f}", flush=True)
