#!/bin/bash
# Round-2 artefacts (run on the GPU box; outputs under gpurun_out/, copy the ones to keep into profiles/):
#   1. the driver's exact bench command,  2. rocprofv3 kernel statistics of a 1 + 3 job run of the same script,
#   3. FETCH_SIZE / WRITE_SIZE of the grouped forward's kernels on the standalone ResNet-101 replay (separate --pmc passes)
# Usage: bash tools/run_final_profile_r02.sh <tag>
V=${1:-vX}; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r02_$V.json 2> gpurun_out/bench_r02_$V.err
echo "bench rc $?"; grep "timed region" gpurun_out/bench_r02_$V.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$V
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$V -o p -- python3 $R/bench.py --steps 3 --warmup 1 \
    --no-cpu-baseline --no-alt-solver --no-phases > $R/gpurun_out/r02_${V}_bench_under_rocprof.json 2> $R/gpurun_out/rocprof_$V.err
cp $(find /tmp/prof_$V -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_${V}_bench_kernel_stats.csv
rm -rf /tmp/prof_$V
grep "timed region" $R/gpurun_out/rocprof_$V.err
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_rn101 fwd_batch_rn101.hip -L$R/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$R/pleas_merging_amd/csrc 2>/dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcf_$c
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcf_$c -o pmc -- /tmp/fwd_rn101 $R/tools/hipbench/rn101_layers.txt 3 > /tmp/pmcf_$c.log 2>&1 || echo "rocprofv3 $c failed"
  f=$(find /tmp/pmcf_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f fwd_batch | tee $R/gpurun_out/r02_pmc_fwd_$c.txt
done
