#!/usr/bin/env bash
# Builds timing-experiment variants of liblrp_hip.so into tools/_ablate/<n>/ (LRP_ABLATE=n
# switches parts of the bicubic window kernel off; outputs are WRONG by design).
# Run a variant with:  LD_LIBRARY_PATH=tools/_ablate/<n> tools/kbench eqd_rect_bc
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/image-lens-reproject_amd/csrc"
# usage: ablate.sh <name>[:-Dflag[,-Dflag...]] ...   (a bare number n means -DLRP_ABLATE=n)
for spec in "$@"; do
  n="${spec%%:*}"; extra=()
  if [[ "$spec" == *:* ]]; then IFS=, read -ra extra <<< "${spec#*:}"; else extra=(-DLRP_ABLATE=$n); fi
  out="$root/tools/_ablate/$n"; mkdir -p "$out"
  FLAGS=(--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
         -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function -I"$src" -I"$root/include" "${extra[@]}")
  pids=()
  for s in lrp_kernels_nn.hip lrp_kernels_bl.hip lrp_kernels_bc.hip lrp_tile_nn.hip lrp_tile_bl.hip lrp_tile_bc.hip lrp_tile_win.hip lrp_tile_winq.hip lrp_tile_win3.hip lrp_tile_winq3.hip lrp_tile_win5.hip lrp_tile_winq5.hip lrp_tile_winy.hip lrp_tile_winx.hip lrp_tile_winy3.hip lrp_tile_winx3.hip lrp_tile_winy5.hip lrp_tile_winx5.hip lrp_tile_winr.hip lrp_tile_winr3.hip lrp_tile_winr5.hip lrp_tables.hip lrp_aux_kernels.hip lrp_pixel_kernels.hip lrp_capi.cpp lrp_host_util.cpp; do
    ( /opt/rocm/bin/hipcc "${FLAGS[@]}" -x hip -c "$src/$s" -o "$out/${s%.*}.o" ) & pids+=($!)
  done
  for p in "${pids[@]}"; do wait "$p"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/liblrp_hip.so" "$out"/*.o
  rm -f "$out"/*.o
  echo "built $out/liblrp_hip.so"
done
