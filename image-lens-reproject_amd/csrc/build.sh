#!/usr/bin/env bash
# Builds liblrp_hip.so (gfx950) in-tree.  Usage: build.sh [jobs]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${LRP_BUILD_OUT:-$here/../lib}"   # (LRP_BUILD_OUT / LRP_BUILD_FLAGS: variant builds for same-box A/B timing, tools/ablate.sh)
obj="$out/obj"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
jobs="${1:-$(( $(nproc) + 4 ))}"
# Parity flags: no FMA contraction, IEEE divide/sqrt, denormals kept, no fast-math.  --offload-compress: the code objects are
# stored compressed in the fat binary (the runtime unpacks them when the library is loaded): liblrp_hip.so 39.7 -> ~8 MB.
FLAGS=(--offload-arch=gfx950 --offload-compress -std=c++17 -O3 -fPIC -ffp-contract=off
       -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-gpu-flush-denormals-to-zero
       -Wall -Wno-unused-function -Wno-inline-asm -Wno-cuda-compat -I"$here" -I"$here/../../include" ${LRP_BUILD_FLAGS:-})
srcs=(lrp_kernels_nn.hip lrp_kernels_bl.hip lrp_kernels_bc.hip lrp_tile_nn.hip lrp_tile_bl.hip lrp_tile_bc.hip lrp_tile_win.hip lrp_tile_winq.hip lrp_tile_win3.hip lrp_tile_winq3.hip lrp_tile_win5.hip lrp_tile_winq5.hip lrp_tile_winy.hip lrp_tile_winx.hip lrp_tile_winy3.hip lrp_tile_winx3.hip lrp_tile_winy5.hip lrp_tile_winx5.hip lrp_tile_winr.hip lrp_tile_winr3.hip lrp_tile_winr5.hip lrp_tile_wing.hip lrp_tile_wing3.hip lrp_tile_wing5.hip lrp_tile_wins.hip lrp_tile_wins3.hip lrp_tile_wins5.hip lrp_tile_winsg.hip lrp_tile_winsg3.hip lrp_tile_winsg5.hip lrp_tile_ssg.hip lrp_tables.hip lrp_geo_lists.hip lrp_aux_kernels.hip lrp_pixel_kernels.hip lrp_capi.cpp lrp_plan.cpp lrp_geocache.cpp lrp_host_util.cpp)
# Per-unit code generation options (measured on MI355X, tools/ablate.sh variants; bits are unaffected):
#   the plain-block window kernels schedule for instruction-level parallelism: rectilinear -> equirectangular bicubic
#   (BASELINE configs[3]) 229 -> 217 us, the other plain-block mappings within +-1 %; the mirrored units gain nothing.
unit_flags() {
  case "$1" in
    lrp_tile_win.hip|lrp_tile_win3.hip|lrp_tile_win5.hip) echo "-mllvm -amdgpu-sched-strategy=max-ilp" ;;
    *) echo "" ;;
  esac
}
pids=()
for s in "${srcs[@]}"; do
  o="$obj/${s%.*}.o"
  if [[ ! -f "$o" || "$here/$s" -nt "$o" || -n "$(find "$here" -maxdepth 1 -name '*.h' -newer "$o" -print -quit)" \
        || "$here/../../include/lrp.h" -nt "$o" || "${BASH_SOURCE[0]}" -nt "$o" ]]; then
    # at most $jobs compilers at once (a window-kernel unit takes 1-2 GiB)
    while (( $(jobs -rp | wc -l) >= jobs )); do wait -n || true; done
    # shellcheck disable=SC2046
    ( "$HIPCC" "${FLAGS[@]}" $(unit_flags "$s") -x hip -c "$here/$s" -o "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
objs=()
for s in "${srcs[@]}"; do objs+=("$obj/${s%.*}.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out/liblrp_hip.so" "${objs[@]}"
echo "built $out/liblrp_hip.so"
