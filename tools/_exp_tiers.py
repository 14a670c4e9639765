# Experiment: blocks per tier of the plain RGBA window kernel (a -DLRP_TIER_STATS build: tools/ablate.sh ts:-DLRP_TIER_STATS),
# rect -> equirect bicubic row bands.  usage: python3 tools/_exp_tiers.py <liblrp_hip.so>
import sys, os, importlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
native = importlib.import_module("image-lens-reproject_amd._native")
native.LIB_PATH = os.path.abspath(sys.argv[1])
lrp = importlib.import_module("image-lens-reproject_amd")
lib = native.load()
n = 4096
lin = lrp.LensInfo.rectilinear(18.0, 36.0, n, n)
lout = lrp.LensInfo.equirectangular()
src = torch.rand((n, n, 4), device="cuda")
dst = torch.empty((n, n, 4), device="cuda")
rot = np.eye(3, dtype=np.float32)
out = (ctypes.c_uint * 8)()
for first, count in [(0, 512), (512, 512), (1024, 512), (1536, 512), (0, 4096)]:
    lib.lrp_debug_read_tiers_plain(out)
    lrp.reproject_rows(lrp.Image(lin, n, n, 4, src), lrp.Image(lout, n, n, 4, dst), 1, 2, first, count, rot)
    torch.cuda.synchronize()
    lib.lrp_debug_read_tiers_plain(out)
    print(f"rows [{first},{first+count}): coef {out[0]} raw {out[1]} direct {out[2]} corner {out[3]} edge-row {out[4]} edge-col {out[5]}", flush=True)
