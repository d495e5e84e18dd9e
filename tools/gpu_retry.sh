#!/bin/bash
# tools/gpu_retry.sh <log> <gpurun timeout s> <command...>: gpurun with retries while every GPU slot of the pod is busy (exit code 3)
LOG=$1; shift; TMO=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$TMO" -- "$@" > "$LOG" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
