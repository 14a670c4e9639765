#!/usr/bin/env bash
# Run on an MI355X box from the repo root: two separate PMC passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), then
# tools/traffic_summary.py turns them into profiles/traffic_r02.json.
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$R/gpurun_out/traffic"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/$ctr" -- python3 "$R/tools/traffic_probe.py" > "$out/$ctr.log" 2>&1
  echo "$ctr pass rc=$?"
done
python3 "$R/tools/traffic_summary.py" "$out" "$R/gpurun_out/traffic_r02.json"
