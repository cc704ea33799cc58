#!/bin/bash
# tools/gpu.sh <timeout-seconds> <log> <command...>: gpurun with retries while no box / slot is free (exit code 3)
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
