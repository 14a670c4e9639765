#!/usr/bin/env python3
"""Where a single launch of the window kernel loses time: start / end of every wavefront (100 MHz wall clock) and the
hardware slot it ran in, from a diagnostic build that records them:

    tools/ablate_units.sh stamps lrp_tile_wing.hip -DLRP_WAVE_STAMPS      # -> tools/_ablate/stamps/liblrp_hip.so
    python3 tools/wave_timeline.py tools/_ablate/stamps/liblrp_hip.so [in_lens out_lens]      (on an MI355X box)

Prints the span of the launch, the lifetimes of its wavefronts, how full the wave slots were over the launch, the ramp
at its start, the tail at its end and the gaps between consecutive wavefronts of one slot."""
import ctypes
import importlib
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

native = importlib.import_module("image-lens-reproject_amd._native")
native.LIB_PATH = os.path.abspath(sys.argv[1])
lrp = importlib.import_module("image-lens-reproject_amd")
lib = native.load()
n = 4096
L = lrp.LensInfo
lens = {"rect": L.rectilinear(18.0, 36.0, n, n), "eqr": L.equirectangular(), "eqd": L.equidistant(math.pi)}
a, b = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("eqd", "rect")
srcs = [torch.rand((n, n, 4), device="cuda") for _ in range(6)]
dsts = [torch.empty((n, n, 4), device="cuda") for _ in range(6)]
rot = lrp.rotation_matrix(0.0, 0.0, 0.0) if a != "eqd" else None
for i in range(8):  # the first launch fills the geometry cache, the others read it
    lrp.reproject(lrp.Image(lens[a], n, n, 4, srcs[i % 6]), lrp.Image(lens[b], n, n, 4, dsts[i % 6]), 1, 2, rot)
torch.cuda.synchronize()
W = 65536
raw = (ctypes.c_ulonglong * (3 * W))()
lib.lrp_debug_read_wave_stamps(raw, W)
v = np.frombuffer(raw, dtype=np.uint64).reshape(W, 3)
v = v[v[:, 0] != 0]
start, end, hw = v[:, 0].astype(np.int64), v[:, 1].astype(np.int64), v[:, 2]
t0 = start.min()
start, end = (start - t0) * 0.01, (end - t0) * 0.01  # us
span = end.max()
life = end - start
print(f"{a} -> {b} 4096^2 RGBA, single launch reading the geometry cache: {len(v)} wavefronts, span {span:.1f} us")
print(f"wavefront lifetime us: mean {life.mean():.2f}  p5 {np.percentile(life, 5):.2f}  median {np.median(life):.2f}  p95 {np.percentile(life, 95):.2f}  max {life.max():.2f}")
simd = ((hw >> 32) & 15) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 16 + ((hw >> 4) & 3)  # xcc, se, sh, cu, simd
n_simd = len(np.unique(simd))
print(f"SIMDs seen {n_simd}; wavefront-time / (span x SIMDs) = {life.sum() / (span * n_simd):.2f} wave slots busy on average")
# occupancy over time
edges = np.linspace(0.0, span, 41)
busy = np.zeros(40)
for k in range(40):
    lo, hi = edges[k], edges[k + 1]
    busy[k] = (np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None)).sum() / ((hi - lo) * n_simd)
print("wave slots busy per SIMD over the launch (40 equal intervals):")
print("  " + " ".join(f"{x:.1f}" for x in busy))
first = np.array([start[simd == s].min() for s in np.unique(simd)])
last = np.array([end[simd == s].max() for s in np.unique(simd)])
print(f"ramp: first wavefront of a SIMD starts at us  p5 {np.percentile(first, 5):.2f} median {np.median(first):.2f} p95 {np.percentile(first, 95):.2f} max {first.max():.2f}")
print(f"tail: last wavefront of a SIMD ends at us     min {last.min():.2f} p5 {np.percentile(last, 5):.2f} median {np.median(last):.2f} p95 {np.percentile(last, 95):.2f} max {last.max():.2f}")
# gaps between consecutive wavefronts of one hardware slot
slot = simd * 16 + (hw & 15)
gaps = []
for s in np.unique(slot):
    m = slot == s
    order = np.argsort(start[m])
    st, en = start[m][order], end[m][order]
    gaps.extend((st[1:] - en[:-1]).tolist())
gaps = np.array(gaps if gaps else [0.0])
blocks = (hw >> 40) & 0xFFFFF
print(f"blocks rendered: total {int(blocks.sum())} (the image has {(n // 16) ** 2}); per wavefront min {int(blocks.min())} median {int(np.median(blocks))} max {int(blocks.max())}")
label = (hw >> 60) & 7
xcc_of = (hw >> 32) & 15
mism = int((label != (xcc_of & 7)).sum())
print(f"wavefronts whose blockIdx.x % 8 differs from the XCD they ran on: {mism}")
print(f"gap between consecutive wavefronts of one slot us: n {len(gaps)} mean {gaps.mean():.2f} median {np.median(gaps):.2f} p95 {np.percentile(gaps, 95):.2f}; sum / (span x SIMDs x 4) = {gaps.sum() / (span * n_simd * 4):.3f}")
xcc = (hw >> 32) & 15
for x in np.unique(xcc):
    m = xcc == x
    print(f"  XCD {int(x)}: {int(m.sum())} wavefronts, first start {start[m].min():.2f}, last end {end[m].max():.2f}, wavefront-time {life[m].sum() / 1e3:.2f} ms")
