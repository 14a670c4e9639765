#!/usr/bin/env bash
# Run on an MI355X box from the repo root: two separate PMC passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), then
# tools/traffic_summary.py turns them into gpurun_out/traffic_<round>.json (copy it to profiles/).  usage: collect_traffic.sh [round: r05]
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
round="${1:-r05}"
out="$R/gpurun_out/traffic"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/$ctr" -- python3 "$R/tools/traffic_probe.py" "$out/order.json" > "$out/$ctr.log" 2>&1
  echo "$ctr pass rc=$?"
done
# cross-check of the read side: the L2's read requests to the fabric by size (no streaming-pattern factor involved)
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$out/RDREQ" -- python3 "$R/tools/traffic_probe.py" "$out/order.json" > "$out/RDREQ.log" 2>&1
echo "RDREQ pass rc=$?"
python3 "$R/tools/traffic_summary.py" "$out" "$R/gpurun_out/traffic_$round.json" > "$out/summary.log" 2>&1; tail -5 "$out/summary.log"
