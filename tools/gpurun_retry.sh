#!/bin/bash
# tools/gpurun_retry.sh <timeout> <command>: gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > /tmp/gpurun_last.log 2>&1; rc=$?
  if [ $rc -ne 3 ]; then cat /tmp/gpurun_last.log; exit $rc; fi
  sleep 45
done
cat /tmp/gpurun_last.log; exit 3
