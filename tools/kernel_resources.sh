#!/usr/bin/env bash
# VGPR / SGPR / spill counts of every kernel of the given translation units (device-only assembly; no GPU needed).
# usage: tools/kernel_resources.sh lrp_tile_winq.hip [more units] [-- extra hipcc flags]
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/image-lens-reproject_amd/csrc"
units=(); extra=()
while [[ $# -gt 0 ]]; do if [[ "$1" == "--" ]]; then shift; extra=("$@"); break; fi; units+=("$1"); shift; done
tmp="$(mktemp -d)"
for u in "${units[@]}"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -fno-fast-math -fno-gpu-flush-denormals-to-zero -I"$src" -I"$root/include" "${extra[@]}" -x hip -c "$src/$u" \
      --cuda-device-only -S -o "$tmp/${u%.*}.s" 2>/dev/null ) &
done
wait
for u in "${units[@]}"; do
  echo "== $u"
  grep -E "^\s+\.name:|\.vgpr_count|\.vgpr_spill_count|\.sgpr_spill_count|\.sgpr_count" "$tmp/${u%.*}.s" | paste - - - - - |
    sed -E 's/\s+/ /g; s/_ZN3lrp[0-9]+//; s/EEvNS_7KParamsE//' |
    awk '{printf "%-60s sgpr %s (spilled %s) vgpr %s (spilled %s)\n", $2, $4, $6, $8, $10}'
done
if [[ -n "${KEEP_ASM:-}" ]]; then echo "assembly kept in $tmp"; else rm -rf "$tmp"; fi
