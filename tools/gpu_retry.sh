#!/bin/bash
# usage: tools/gpu_retry.sh <log> <timeout> <cmd>   -- retries while no slot is free (exit 3)
log=$1; to=$2; shift 2
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
