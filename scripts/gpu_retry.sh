#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit 3 / "transient"): usage scripts/gpu_retry.sh <timeout s> '<command>'
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  out=$(/usr/local/graft/bin/gpurun --timeout $t -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out"; exit $rc
done
echo "$out"; exit 3
