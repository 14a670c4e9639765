# Experiment: where does rect -> equirect bicubic spend its time?  Row bands of the output, single launches.
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
lrp = importlib.import_module("image-lens-reproject_amd")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 4096
lin = lrp.LensInfo.rectilinear(18.0, 36.0, n, n)
lout = lrp.LensInfo.equirectangular()
srcs = [torch.rand((n, n, C), device="cuda") for _ in range(8)]
dsts = [torch.empty((n, n, C), device="cuda") for _ in range(8)]
rot = np.eye(3, dtype=np.float32)
def run(first, count, reps=24):
    for i in range(4):
        lrp.reproject_rows(lrp.Image(lin, n, n, C, srcs[i % 8]), lrp.Image(lout, n, n, C, dsts[i % 8]), 1, 2, first, count, rot)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        lrp.reproject_rows(lrp.Image(lin, n, n, C, srcs[i % 8]), lrp.Image(lout, n, n, C, dsts[i % 8]), 1, 2, first, count, rot)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for first, count in [(0, 4096), (0, 1024), (1024, 2048), (1024, 1024), (1536, 1024), (2048, 512), (3072, 1024), (0, 512), (512, 512), (1024, 512), (1536, 512)]:
    print(f"C={C} rows [{first},{first+count}): {run(first, count):8.1f} us", flush=True)
