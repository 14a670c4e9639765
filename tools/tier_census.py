#!/usr/bin/env python3
"""Blocks per tier of the window kernel (coefficient planes / raw taps / per-pixel gathers / corner / edge / split), per
mapping — the census DESIGN.md sections 4.1 and 5 quote.  Needs a diagnostic build of the library that counts them:

    tools/ablate.sh ts:-DLRP_TIER_STATS          # -> tools/_ablate/ts/liblrp_hip.so
    python3 tools/tier_census.py tools/_ablate/ts/liblrp_hip.so        (on an MI355X box)

The counters live in the plain-block RGBA unit (lrp_tile_win.hip), so the mirror modes are switched off for the run; the
tier of a block does not depend on the mode.  The cubemap faces (8192^2 -> 2048^2) are rendered as RGBA stand-ins."""
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
lrp.debug_set("mirror_modes", 0)
lrp.debug_set("quad", 0)
NAMES = ["coefficient", "raw", "direct", "corner", "edge-row", "edge-col", "split"]
out = (ctypes.c_uint * 8)()


def rot(deg):
    p, t, r = [float(np.float32(d) * np.float32(math.pi) / np.float32(180.0)) for d in deg]
    return lrp.rotation_matrix(p, t, r)


def census(what, lin, n_in, lout, n_out, deg):
    src = torch.rand((n_in, n_in, 4), device="cuda")
    dst = torch.empty((n_out, n_out, 4), device="cuda")
    lib.lrp_debug_read_tiers_plain(out)  # reset
    lrp.reproject(lrp.Image(lin, n_in, n_in, 4, src), lrp.Image(lout, n_out, n_out, 4, dst), 1, 2, rot(deg))
    torch.cuda.synchronize()
    lib.lrp_debug_read_tiers_plain(out)
    tot = sum(out[:7]) or 1
    print(f"{what:44s}" + "  ".join(f"{nm} {100.0 * out[i] / tot:5.1f}%" for i, nm in enumerate(NAMES)), flush=True)


L = lrp.LensInfo
n = 4096
lens = {"rect": L.rectilinear(18.0, 36.0, n, n), "eqr": L.equirectangular(), "eqd": L.equidistant(math.pi)}
for a, b, deg in [("eqd", "rect", (0, 0, 0)), ("eqr", "rect", (0, 0, 0)), ("eqr", "rect", (30, -15, 5)), ("eqr", "rect", (0, 90, 0)),
                  ("eqr", "eqd", (30, -15, 5)), ("eqd", "eqd", (10, 5, 0)), ("eqr", "eqr", (30, -15, 5)), ("rect", "rect", (10, 5, 0)),
                  ("rect", "eqd", (0, 0, 0)), ("rect", "eqr", (0, 0, 0))]:
    census(f"4096^2 {a} -> {b} rot={deg}", lens[a], n, lens[b], n, deg)
for deg in [(0, 0, 0), (90, 0, 0), (0, 90, 0)]:
    census(f"8192^2 eqr -> 2048^2 rect rot={deg}", L.equirectangular(), 8192, L.rectilinear(18.0, 36.0, 2048, 2048), 2048, deg)
