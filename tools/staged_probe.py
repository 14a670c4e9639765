#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) throughput of the batch path (lrp_context_*), for DESIGN.md.
Pageable numpy frames in, pageable numpy frames out, 4096^2 RGBA, fisheye->rect bicubic."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
pkg = importlib.import_module("image-lens-reproject_amd")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e: print("cpu.max n/a", e)
size, c, n = 4096, 4, 8
wl = bench.WORKLOADS["fisheye_to_rect_bicubic"]
lin, lout = bench.make_lens(pkg, wl["in_lens"], size, size), bench.make_lens(pkg, wl["out_lens"], size, size)
d = torch.empty((size, size, c), dtype=torch.float32, device="cuda"); pkg.synth_fill(d, size, size, c, 0x5EED0000); torch.cuda.synchronize()
src = d.cpu().numpy()
for pinned in (False, True):
    if pinned:
        ins = [torch.from_numpy(src).clone().pin_memory().numpy() for _ in range(n)]
        outs = [torch.empty((size, size, c), dtype=torch.float32).pin_memory().numpy() for _ in range(n)]
    else:
        ins = [src.copy() for _ in range(n)]; outs = [np.empty_like(src) for _ in range(n)]
    for streams in (1, 3):
        with pkg.BatchContext(device=0, n_streams=streams) as ctx:
            for rep in range(2):
                t0 = time.perf_counter()
                for i, o in zip(ins, outs):
                    ctx.submit(pkg.Image(lin, size, size, c, i), pkg.Image(lout, size, size, c, o), 1, 2, None)
                ctx.wait()
                dt = time.perf_counter() - t0
            print(f"pinned={pinned} streams={streams}: {n*size*size/dt/1e6:.0f} Mpix/s staged ({dt/n*1e3:.1f} ms/frame, {n*2*size*size*c*4/dt/1e9:.1f} GB/s over PCIe both ways)")
