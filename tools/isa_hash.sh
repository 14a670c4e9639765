#!/usr/bin/env bash
# sha256 of the gfx950 machine code of every kernel object in a directory (one line per object): two builds whose sources differ
# only in dead alternatives print the same lines.  usage: isa_hash.sh <obj-dir>
set -euo pipefail
B=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d); trap 'rm -rf $tmp' EXIT
for o in "$1"/*.o; do
  n=$(basename "$o")
  if $B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$o" /dev/null 2>/dev/null && [ -s $tmp/fat.bin ]; then
    $B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle 2>/dev/null
    echo "$n $($B/llvm-objdump -d $tmp/dev.co | grep -v "file format" | sha256sum | cut -c1-16)"
  else
    echo "$n host-only $(sha256sum < "$o" | cut -c1-16)"
  fi
  rm -f $tmp/fat.bin $tmp/dev.co
done
