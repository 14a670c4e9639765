#!/usr/bin/env python3
"""The automatic choices of enqueue_reproject (csrc/lrp_capi.cpp) against their alternatives, on THIS box: for every policy switch
that picks a kernel path by a measured rule — block lists, the big-window variant, strip length, frames per wavefront, mirror
modes against the geometry cache, merged multi-output launches, the supersampling instantiations, the fused corner fill — the
workloads the rule was measured on are timed with the default and with every other setting of the switch, inside one process
("thresholds carry single-box A/B numbers in comments and no test that the chosen path is the fast
one").  Prints one line per (workload, switch) and FAILS (exit 1) when a default is more than `tolerance` slower than the best
alternative.  usage: policy_check.py [tolerance: 0.05]"""
import importlib
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

pkg = importlib.import_module("image-lens-reproject_amd")
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
dev = torch.device("cuda", 0)
N_RES = 16
_frames = {}


def frames(size, c, out_size):
    key = (size, c, out_size)
    if key not in _frames:
        _frames.clear()
        torch.cuda.empty_cache()
        srcs, dsts = [], []
        for k in range(N_RES if size <= 4096 else 4):
            s = torch.empty((size, size, c), dtype=torch.float32, device=dev)
            pkg.synth_fill(s, size, size, c, 0x5EED0000 + k, 4 if c == 5 else -1)
            srcs.append(s)
            dsts.append(torch.empty((out_size, out_size, c), dtype=torch.float32, device=dev))
        _frames[key] = (srcs, dsts)
    return _frames[key]


def lens(kind, n):
    L = pkg.LensInfo
    return {"rect": L.rectilinear(18.0, 36.0, n, n), "eqd": L.equidistant(3.14159265), "eqr": L.equirectangular()}[kind]


def rot(deg):
    return None if deg is None else pkg.rotation_matrix(*[d * math.pi / 180.0 for d in deg])


def time_us(wl, batch, reps):
    size, out_size, c, ns = wl.get("size", 4096), wl.get("out_size", wl.get("size", 4096)), wl.get("c", 4), wl.get("ns", 1)
    srcs, dsts = frames(size, c, out_size)
    lin, lout = lens(wl["inp"], size), lens(wl["out"], out_size)
    ins = [pkg.Image(lin, size, size, c, s) for s in srcs]
    outs = [pkg.Image(lout, out_size, out_size, c, d) for d in dsts]
    R, post = rot(wl.get("deg")), wl.get("post")
    n = len(srcs)

    def launch(i):
        if batch:
            pkg.reproject_batch(ins[:batch], outs[:batch], ns, wl["interp"], R, post=post)
        else:
            pkg.reproject(ins[i % n], outs[i % n], ns, wl["interp"], R, post=post)

    for i in range(4):  # the launch that fills the cache, the lists, then the steady state
        launch(i)
        torch.cuda.synchronize()
    for i in range(12):
        launch(i)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps / (batch or 1)
        best = us if best is None else min(best, us)
    return best


HEAD = dict(inp="eqd", out="rect", interp=2, deg=None)
GEN = dict(inp="eqr", out="rect", interp=2, deg=(30.0, -15.0, 5.0))
C3 = dict(inp="rect", out="eqr", interp=2, deg=(0.0, 0.0, 0.0))
C3Z = dict(C3, c=5, post=(2.0, 4.0))
RFE = dict(inp="rect", out="eqd", interp=2, deg=None)
BL = dict(inp="eqr", out="eqd", interp=1, deg=(30.0, -15.0, 5.0))
NN = dict(inp="eqr", out="rect", interp=0, deg=(0.0, 0.0, 0.0))
SS = dict(HEAD, out_size=2048, ns=2)
CHECKS = [  # (what, workload, batch, switch, settings: the first is the default)
    ("block lists, configs[3] RGBA single", C3, 0, "geo_lists", (1, 0, 2)),
    ("block lists, configs[3] RGBAZ + tonemap single", C3Z, 0, "geo_lists", (1, 0, 2)),
    ("block lists, configs[3] RGBAZ + tonemap batch16", C3Z, 16, "geo_lists", (1, 0, 2)),
    ("block lists, rect -> fisheye single", RFE, 0, "geo_lists", (1, 0, 2)),
    ("corner fill as shares, configs[3] RGBAZ + tonemap single", C3Z, 0, "geo_fill_fused", (1, 0)),
    ("big-window variant, configs[3] RGBA single", C3, 0, "geo_big", (1, 0)),
    ("big-window variant by census, rect -> fisheye RGBA single", RFE, 0, "geo_big", (1, 0, 2)),
    ("big-window variant by census, rect -> fisheye RGBA batch16", RFE, 16, "geo_big", (1, 0, 2)),
    ("big-window variant by census, general rotation single (no wide block: stays out)", GEN, 0, "geo_big", (1, 0, 2)),
    ("big-window variant by census, 8192^2 -> 2048^2 pole face RGB", dict(inp="eqr", out="rect", interp=2, deg=(0.0, 90.0, 0.0), size=8192, out_size=2048, c=3), 0, "geo_big", (1, 0, 2)),
    ("big-window variant by census, 8192^2 -> 2048^2 side face RGB", dict(inp="eqr", out="rect", interp=2, deg=(90.0, 0.0, 0.0), size=8192, out_size=2048, c=3), 0, "geo_big", (1, 0, 2)),
    ("big-window variant by census, 8192^2 -> 2048^2 face pitched 45 degrees RGB", dict(inp="eqr", out="rect", interp=2, deg=(0.0, 45.0, 0.0), size=8192, out_size=2048, c=3), 0, "geo_big", (1, 0, 2)),
    ("big-window variant by census, 8192^2 -> 2048^2 face pitched 65 degrees RGB", dict(inp="eqr", out="rect", interp=2, deg=(20.0, 65.0, 0.0), size=8192, out_size=2048, c=3), 0, "geo_big", (1, 0, 2)),
    ("tap DMA, configs[3] RGBA single", C3, 0, "win_tapdma", (1, 0)),
    ("tap DMA, configs[3] RGBAZ + tonemap batch16", C3Z, 16, "win_tapdma", (1, 0)),
    ("strip length of reading launches, headline single", HEAD, 0, "geo_strip", (0, 1, 2, 4)),
    ("strip length of reading launches, general rotation single", GEN, 0, "geo_strip", (0, 1, 2, 4)),
    ("frames per wavefront, headline batch16", HEAD, 16, "batch_frames", (0, 1, 4, 16)),
    ("frames per wavefront, configs[3] RGBA batch16", C3, 16, "batch_frames", (0, 1, 16)),
    ("geometry cache, headline single", HEAD, 0, "geo_cache", (1, 0)),
    ("geometry cache, general rotation single", GEN, 0, "geo_cache", (1, 0)),
    ("geometry cache, bilinear rotated single", BL, 0, "geo_cache", (1, 0)),
    ("geometry cache, bilinear rotated batch16", BL, 16, "geo_cache", (1, 0)),
    ("geometry cache, nearest without a rotation single", NN, 0, "geo_cache", (1, 0)),
    ("supersampling instantiations, 4096^2 -> 2048^2 num_samples 2 single", SS, 0, "win_ss", (1, 0)),
    ("supersampling instantiations, 4096^2 -> 1365^2 num_samples 3 single", dict(HEAD, out_size=1365, ns=3), 0, "win_ss", (1, 0)),
    ("supersampling instantiations, 4096^2 -> 1024^2 num_samples 4 single", dict(HEAD, out_size=1024, ns=4), 0, "win_ss", (1, 0)),
    ("entry of sub-samples, 4096^2 -> 2048^2 num_samples 2 single", SS, 0, "geo_cache", (1, 0)),
    ("entry of sub-samples, general rotation 4096^2 -> 1024^2 num_samples 4 single", dict(GEN, out_size=1024, ns=4), 0, "geo_cache", (1, 0)),
    ("entry of sub-samples, bilinear rotated into a fisheye frame 4096^2 -> 2048^2 num_samples 2 single", dict(BL, out_size=2048, ns=2), 0, "geo_cache", (1, 0)),
    ("entry of sub-samples, bilinear rotated into a fisheye frame 4096^2 -> 1024^2 num_samples 4 single", dict(BL, out_size=1024, ns=4), 0, "geo_cache", (1, 0)),
    ("entry of sub-samples, equirect -> rect bilinear rotated 4096^2 -> 1365^2 num_samples 3 single", dict(GEN, interp=1, out_size=1365, ns=3), 0, "geo_cache", (1, 0)),
]
bad = 0
print(f"# default against the alternatives of every policy switch, us per frame (tolerance {tol:.0%}); box: {torch.cuda.get_device_name(0)}")
for what, wl, batch, switch, settings in CHECKS:
    times = {}
    for v in settings + (settings[0],):  # (the default first AND last: the better of the two, so that drift over the run does not decide)
        prev = pkg.debug_set(switch, v)
        pkg.release_cached_tables()
        t = time_us(wl, batch, 6 if batch else 24)
        times[v] = min(t, times.get(v, t))
        pkg.debug_set(switch, prev)
    default, best = times[settings[0]], min(times.values())
    ok = default <= best * (1.0 + tol)
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'SLOW'} {what:70s} {switch}: " + "  ".join(f"{v}: {t:7.1f}" for v, t in times.items()) + f"   default / best = {default / best:.3f}", flush=True)
print(f"# {bad} of {len(CHECKS)} defaults more than {tol:.0%} behind their best alternative")
sys.exit(1 if bad else 0)
