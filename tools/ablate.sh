#!/usr/bin/env bash
# Builds variants of liblrp_hip.so into tools/_ablate/<name>/ for same-box A/B timing: sizes (-DLRP_WIN_STRIP=4,
# -DLRP_SS_STRIP=2, -DLRP_WIN_CAP_BIG=1280 ...) and compiler options (-mllvm ...), through csrc/build.sh (same units, same
# per-unit options).  usage: ablate.sh <name>:<flag>[,<flag>...] ...
# Run a variant with:  LD_LIBRARY_PATH=tools/_ablate/<name> tools/kbench eqd_rect_bc
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for spec in "$@"; do
  n="${spec%%:*}"; flags=""
  if [[ "$spec" == *:* ]]; then flags="$(echo "${spec#*:}" | tr ',' ' ')"; fi
  LRP_BUILD_OUT="$root/tools/_ablate/$n" LRP_BUILD_FLAGS="$flags" bash "$root/image-lens-reproject_amd/csrc/build.sh"
  rm -rf "$root/tools/_ablate/$n/obj"
done
