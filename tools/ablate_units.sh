#!/usr/bin/env bash
# Variant of liblrp_hip.so for same-box A/B timing in which only SOME translation units are recompiled with other options:
#   tools/ablate_units.sh <name> "<unit.hip> ..." [hipcc options ...]
# builds tools/_ablate/<name>/liblrp_hip.so from those units (compiled here, WITHOUT the per-unit options of csrc/build.sh
# unless given) and the up-to-date objects of image-lens-reproject_amd/lib/obj.  Run with
#   LD_LIBRARY_PATH=tools/_ablate/<name> tools/kbench ...
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/image-lens-reproject_amd/csrc"; obj="$root/image-lens-reproject_amd/lib/obj"
name="$1"; units="$2"; shift 2
out="$root/tools/_ablate/$name"; rm -rf "$out"; mkdir -p "$out"
FLAGS=(--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
       -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function -Wno-inline-asm -I"$src" -I"$root/include" "$@")
pids=()
for s in $units; do ( /opt/rocm/bin/hipcc "${FLAGS[@]}" -x hip -c "$src/$s" -o "$out/${s%.*}.o" ) & pids+=($!); done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for o in "$obj"/*.o; do b="$(basename "$o")"; [[ -f "$out/$b" ]] && objs+=("$out/$b") || objs+=("$o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/liblrp_hip.so" "${objs[@]}"
rm -f "$out"/*.o
echo "built $out/liblrp_hip.so"
