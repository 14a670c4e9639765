#!/usr/bin/env bash
# Builds tools/kbench (native kernel bench over the C ABI) against the in-tree liblrp_hip.so.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
lib="$root/image-lens-reproject_amd/lib"
/opt/rocm/bin/hipcc -O2 -std=c++17 "$root/tools/kbench.cpp" -I"$root/include" -L"$lib" -llrp_hip \
  -Wl,-rpath,'$ORIGIN/../image-lens-reproject_amd/lib' -ldl -o "$root/tools/kbench"
echo "built $root/tools/kbench"
/opt/rocm/bin/hipcc -O2 -std=c++17 "$root/tools/staged_bench.cpp" -I"$root/include" -L"$lib" -llrp_hip \
  -Wl,-rpath,'$ORIGIN/../image-lens-reproject_amd/lib' -o "$root/tools/staged_bench"
echo "built $root/tools/staged_bench"
