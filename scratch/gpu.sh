#!/bin/bash
# scratch/gpu.sh TIMEOUT 'command' : gpurun with retries while the pod's GPU slots are busy (status=transient, nothing charged)
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  OUT=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
  if echo "$OUT" | grep -q "status=transient"; then sleep 60; continue; fi
  echo "$OUT"; exit 0
done
echo "$OUT"; exit 3
