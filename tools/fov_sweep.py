#!/usr/bin/env python3
"""A rectilinear view (focal length f, 36 mm sensor) rendered into a 4096^2 panorama, bicubic, single launches, for several
focal lengths: how the frame time of BASELINE configs[3] splits into its out-of-view part (f = 400: nearly every block a
corner block) and its in-view part (f = 18 is the config).  usage: fov_sweep.py [channels [post]]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
pkg = importlib.import_module("image-lens-reproject_amd")
dev = torch.device("cuda", 0)
size, c, n = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 12
post = (1.0009765625, 4.0) if c == 5 and len(sys.argv) > 2 else None
srcs = [torch.empty((size, size, c), dtype=torch.float32, device=dev) for _ in range(n)]
for k, s in enumerate(srcs):
    pkg.synth_fill(s, size, size, c, 0x5EED0000 + k, 4 if c == 5 else -1)
dsts = [torch.empty((size, size, c), dtype=torch.float32, device=dev) for _ in range(n)]
rot = pkg.rotation_matrix(0.0, 0.0, 0.0)
for f in (18.0, 9.0, 12.0, 27.0, 36.0, 72.0, 400.0):
    lin = pkg.LensInfo.rectilinear(f, 36.0, size, size)
    lout = pkg.LensInfo.equirectangular()
    ins = [pkg.Image(lin, size, size, c, s) for s in srcs]
    outs = [pkg.Image(lout, size, size, c, d) for d in dsts]
    for i in range(4):
        pkg.reproject(ins[i % n], outs[i % n], 1, 2, rot, post=post)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 36
    e0.record()
    for i in range(reps):
        pkg.reproject(ins[i % n], outs[i % n], 1, 2, rot, post=post)
    e1.record()
    torch.cuda.synchronize()
    print(f"focal {f:6.1f}  hfov {2*np.degrees(np.arctan(18.0/f)):6.1f} deg  {e0.elapsed_time(e1)*1e3/reps:7.1f} us per frame")
