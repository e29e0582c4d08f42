#!/bin/bash
# Kernel statistics of the FIRST bench process on a fresh box vs the second (vendor-library caches warm): which kernels differ?
set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for run in first second; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$run -o p -- python3 $R/bench.py --no-cpu-baseline --steps 125 --phase-log \
      > $R/gpurun_out/fs_$run.json 2> $R/gpurun_out/fs_$run.err
  cp $(find /tmp/prof_$run -name "*kernel_stats.csv" | head -1) $R/gpurun_out/fs_${run}_kernel_stats.csv
  rm -rf /tmp/prof_$run
  grep "phase\|timed" $R/gpurun_out/fs_$run.err
done
