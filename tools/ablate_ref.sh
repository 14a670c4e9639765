#!/usr/bin/env bash
# Builds liblrp_hip.so of a git revision into tools/_ablate/<name>/ for same-box A/B timing:
#   tools/ablate_ref.sh <name> <git-ref>;  LD_LIBRARY_PATH=tools/_ablate/<name> tools/kbench ...
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name="$1"; ref="$2"
tmp="$(mktemp -d)"; trap 'rm -rf "$tmp"' EXIT
git -C "$root" archive "$ref" image-lens-reproject_amd/csrc include | tar -x -C "$tmp"
bash "$tmp/image-lens-reproject_amd/csrc/build.sh" > /dev/null
mkdir -p "$root/tools/_ablate/$name"
cp "$tmp/image-lens-reproject_amd/lib/liblrp_hip.so" "$root/tools/_ablate/$name/"
echo "built tools/_ablate/$name/liblrp_hip.so from $ref"
