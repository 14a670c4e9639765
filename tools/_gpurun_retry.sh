#!/usr/bin/env bash
# usage: _gpurun_retry.sh <timeout> '<command>'   — retries while the pod's GPU slots are busy
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > /tmp/gpurun_last.log 2>&1
  if grep -q "status=transient" /tmp/gpurun_last.log; then sleep 60; else break; fi
done
cat /tmp/gpurun_last.log
