# Experiment: blocks per tier for the cubemap faces (8192^2 panorama -> 2048^2 rectilinear, RGBA stand-in, plain kernel)
import sys, os, importlib, ctypes, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LRP_MIRROR_MODES"] = "0"
os.environ["LRP_QUAD"] = "0"
import torch, numpy as np
native = importlib.import_module("image-lens-reproject_amd._native")
native.LIB_PATH = os.path.abspath(sys.argv[1])
lrp = importlib.import_module("image-lens-reproject_amd")
lib = native.load()
n, m = 8192, 2048
L = lrp.LensInfo
src = torch.rand((n, n, 4), device="cuda")
dst = torch.empty((m, m, 4), device="cuda")
out = (ctypes.c_uint * 8)()
def rot(deg):
    p, t, r = [float(np.float32(d) * np.float32(math.pi) / np.float32(180.0)) for d in deg]
    return lrp.rotation_matrix(p, t, r)
for deg in [(0, 0, 0), (90, 0, 0), (0, 90, 0), (0, -90, 0)]:
    for first, count in [(0, 2048), (0, 512), (512, 512), (768, 256), (1024, 256)]:
        lib.lrp_debug_read_tiers_plain(out)
        lrp.reproject_rows(lrp.Image(L.equirectangular(), n, n, 4, src), lrp.Image(L.rectilinear(18.0, 36.0, m, m), m, m, 4, dst), 1, 2, first, count, rot(deg))
        torch.cuda.synchronize()
        lib.lrp_debug_read_tiers_plain(out)
        tot = sum(out[:7]) or 1
        print(f"rot={deg} rows [{first},{first+count}): " + " ".join(f"{nm} {100.0 * out[i] / tot:.1f}%" for i, nm in enumerate(["coef", "raw", "direct", "corner", "edge-row", "edge-col", "split"])), flush=True)
