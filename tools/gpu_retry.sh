#!/bin/bash
# gpurun wrapper: retries ONLY when no slot / box was free (exit code 3: nothing ran, nothing charged).
# usage: tools/gpu_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[gpu_retry] no slot (attempt $attempt), sleeping 120 s" >&2
  sleep 120
done
exit 3
