#!/bin/bash
# Final artefacts of a version: the bench line (with the CPU baseline) and the rocprofv3 kernel statistics of the same command.
# Usage (GPU box): bash tools/run_final_profile.sh v22
set -e
V=${1:-vX}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout -k 10 600 python3 $R/bench.py > $R/gpurun_out/bench_r01_$V.json 2> $R/gpurun_out/bench_r01_$V.err
grep "timed region" $R/gpurun_out/bench_r01_$V.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$V -o p -- python3 $R/bench.py --no-cpu-baseline \
    > $R/gpurun_out/r01_${V}_bench_under_rocprof.json 2> $R/gpurun_out/rocprof_$V.err
cp $(find /tmp/prof_$V -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_${V}_bench_kernel_stats.csv
rm -rf /tmp/prof_$V
grep "timed region" $R/gpurun_out/rocprof_$V.err
