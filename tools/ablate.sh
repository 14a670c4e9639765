#!/usr/bin/env bash
# Builds variants of liblrp_hip.so into tools/_ablate/<name>/ for same-box A/B timing: tuning knobs (-DLRP_WIN_STRIP=4,
# -DLRP_VERT_STEPS=0, -DLRP_NO_PACKED=1 ...), compiler options (-mllvm ...), and the timing-only experiments whose
# outputs are WRONG by design (-DLRP_ABLATE_ONETAP, -DLRP_ABLATE_L2ROWS, -DLRP_NO_STORE, -DLRP_SKIP_PLANES,
# -DLRP_SKIP_TAP_READS, -DLRP_NO_DMA_WAIT).  Note: the per-unit options of csrc/build.sh are not applied here.
# Run a variant with:  LD_LIBRARY_PATH=tools/_ablate/<name> tools/kbench eqd_rect_bc
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/image-lens-reproject_amd/csrc"
# usage: ablate.sh <name>:<flag>[,<flag>...] ...
for spec in "$@"; do
  n="${spec%%:*}"; extra=()
  if [[ "$spec" == *:* ]]; then IFS=, read -ra extra <<< "${spec#*:}"; fi
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
