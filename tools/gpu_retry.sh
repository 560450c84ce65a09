#!/bin/bash
# tools/gpu_retry.sh <log> <timeout> <command...>: gpurun, retried while no box / slot is free (exit code 3)
log="$1"; to="$2"; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
