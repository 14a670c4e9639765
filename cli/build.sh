#!/usr/bin/env bash
# Builds the CLI front end -> image-lens-reproject_amd/bin/reproject (host C++17; links
# liblrp_hip.so, the system libpng16 runtime and zlib; png.h comes from PNG_INCLUDE
# because the image ships the libpng runtime without its development header; jpeglib.h comes
# from the same directory, the libjpeg 9 runtime is opened with dlopen: cli/lrp_jpeg.cpp).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
lib="$root/image-lens-reproject_amd/lib"; out="$root/image-lens-reproject_amd/bin"; mkdir -p "$out"
PNG_INCLUDE="${PNG_INCLUDE:-/opt/conda/include}"
PNG_LIB="${PNG_LIB:-/usr/lib/x86_64-linux-gnu/libpng16.so.16}"
g++ -std=c++17 -O2 -Wall -pthread -I"$root/include" -I"$root/cli" -idirafter "$PNG_INCLUDE" \
  "$root/cli/reproject_main.cpp" "$root/cli/lrp_cli_options.cpp" "$root/cli/lrp_run_plan.cpp" "$root/cli/lrp_engine.cpp" "$root/cli/lrp_image_io.cpp" "$root/cli/lrp_jpeg.cpp" "$root/cli/lrp_config.cpp" \
  -L"$lib" -llrp_hip "$PNG_LIB" -lz -ldl -Wl,-rpath,'$ORIGIN/../lib' -o "$out/reproject"
echo "built $out/reproject"
