# Experiment: blocks per tier of the plain RGBA window kernel for whole-frame single launches of a few mappings
# (a -DLRP_TIER_STATS build).  usage: python3 tools/_exp_tiers2.py <liblrp_hip.so>
import sys, os, importlib, ctypes, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LRP_MIRROR_MODES"] = "0"
os.environ["LRP_QUAD"] = "0"
import torch, numpy as np
native = importlib.import_module("image-lens-reproject_amd._native")
native.LIB_PATH = os.path.abspath(sys.argv[1])
lrp = importlib.import_module("image-lens-reproject_amd")
lib = native.load()
n = 4096
L = lrp.LensInfo
lenses = {"rect": L.rectilinear(18.0, 36.0, n, n), "eqr": L.equirectangular(), "eqd": L.equidistant(math.pi)}
src = torch.rand((n, n, 4), device="cuda")
dst = torch.empty((n, n, 4), device="cuda")
out = (ctypes.c_uint * 8)()
def rot(deg):
    p, t, r = [float(np.float32(d) * np.float32(math.pi) / np.float32(180.0)) for d in deg]
    return lrp.rotation_matrix(p, t, r)
for a, b, deg in [("eqr", "eqd", (30, -15, 5)), ("eqd", "eqd", (10, 5, 0)), ("eqr", "eqr", (30, -15, 5)), ("rect", "rect", (10, 5, 0)),
                  ("rect", "eqd", (0, 0, 0)), ("eqr", "rect", (30, -15, 5)), ("eqr", "rect", (0, 90, 0)), ("eqd", "rect", (0, 0, 0))]:
    lib.lrp_debug_read_tiers_plain(out)
    lrp.reproject(lrp.Image(lenses[a], n, n, 4, src), lrp.Image(lenses[b], n, n, 4, dst), 1, 2, rot(deg))
    torch.cuda.synchronize()
    lib.lrp_debug_read_tiers_plain(out)
    tot = sum(out[:7]) or 1
    print(f"{a}->{b} rot={deg}: " + " ".join(f"{nm} {100.0 * out[i] / tot:.1f}%" for i, nm in enumerate(["coef", "raw", "direct", "corner", "edge-row", "edge-col", "split"])), flush=True)
