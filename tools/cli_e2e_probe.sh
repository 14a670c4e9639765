#!/usr/bin/env bash
# End-to-end CLI timing on an MI355X box (for DESIGN.md): 4096^2 RGBAZ OpenEXR (ZIP) and 4096^2 PNG
# inputs, decode -> GPU reproject (+ tonemap) -> encode, -j 16.
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"
W=/tmp/lrp_e2e; rm -rf $W; mkdir -p $W/exr $W/png $W/out
python3 - <<PY
import sys, time, numpy as np
sys.path.insert(0, "$R/tests")
import exr_util
from PIL import Image
rng = np.random.default_rng(0)
n = 4096
base = (rng.random((n, n)) * 2).astype(np.float16)
t = time.time()
exr_util.write_exr("$W/exr/frame_0000.exr", {c: np.roll(base, i * 37, axis=1) for i, c in enumerate("RGBAZ")}, 3)
print("python exr writer: %.1f s" % (time.time() - t))
Image.fromarray(rng.integers(0, 256, size=(n, n, 3), dtype=np.uint8), "RGB").save("$W/png/frame_0000.png")
PY
for i in 1 2 3 4 5 6 7; do cp $W/exr/frame_0000.exr $W/exr/frame_000$i.exr; cp $W/png/frame_0000.png $W/png/frame_000$i.png; done
CLI=./image-lens-reproject_amd/bin/reproject
for j in 1 8 16; do
  rm -rf $W/out; s=$(date +%s.%N)
  $CLI -i $W/exr -o $W/out --exr --no-configs 4096,4096 --i-rectilinear 18,36 --equirectangular full --bc --exposure 1 --reinhard 4 -j $j > $W/log_exr_$j.txt 2>&1
  e=$(date +%s.%N); echo "EXR RGBAZ 8 frames, -j $j: $(python3 -c "print(round($e-$s,2))") s"
done
for j in 1 8 16; do
  rm -rf $W/out; s=$(date +%s.%N)
  $CLI -i $W/png -o $W/out --png --no-configs 4096,4096 --i-equirectangular full --rectilinear 18,36 --bc -j $j > $W/log_png_$j.txt 2>&1
  e=$(date +%s.%N); echo "PNG RGB 8 frames, -j $j: $(python3 -c "print(round($e-$s,2))") s"
done
tail -2 $W/log_exr_16.txt
