#!/bin/bash
# Evidence for the closed-form kernels (neq_batch_kernel, neq_reduce_kernel, potrf_diag, trsm_rows, trail_update) on the
# ResNet-101 layer list at batch 16, standalone replay (tools/hipbench/neq_batch_rn101.hip):
#   1. plain run (HIP events),  2. rocprofv3 --kernel-trace --stats,  3. FETCH_SIZE and WRITE_SIZE in separate --pmc passes,
#   4. SQ counters + GRBM_GUI_ACTIVE.  Outputs: gpurun_out/r03_neq_<tag>_*.  Usage: bash tools/run_neq_profile_r03.sh <tag>
V=${1:-v0}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/neq_rn101 neq_batch_rn101.hip -L$R/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$R/pleas_merging_amd/csrc 2>/dev/null || { echo "build failed"; exit 1; }
L=$R/tools/hipbench/rn101_layers.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 /tmp/neq_rn101 $L 10 1 | tee $O/r03_neq_${V}_plain.txt || exit 1
rm -rf /tmp/neqprof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/neqprof -o p -- /tmp/neq_rn101 $L 10 1 > /tmp/neqprof.log 2>&1 || { echo "rocprofv3 stats failed"; tail -5 /tmp/neqprof.log; exit 1; }
cp $(find /tmp/neqprof -name "*kernel_stats.csv" | head -1) $O/r03_neq_${V}_kernel_stats.csv
cat $O/r03_neq_${V}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcn_$c
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcn_$c -o pmc -- /tmp/neq_rn101 $L 3 1 > /tmp/pmcn_$c.log 2>&1 || { echo "rocprofv3 $c failed"; tail -5 /tmp/pmcn_$c.log; exit 1; }
  f=$(find /tmp/pmcn_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f neq_ potrf trsm trail ridge | tee $O/r03_neq_${V}_pmc_$c.txt
done
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
rm -rf /tmp/pmcn_sq
timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcn_sq -o pmc -- /tmp/neq_rn101 $L 3 1 > /tmp/pmcn_sq.log 2>&1 || { echo "rocprofv3 SQ failed"; tail -5 /tmp/pmcn_sq.log; exit 1; }
f=$(find /tmp/pmcn_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f neq_ potrf trsm trail | tee $O/r03_neq_${V}_pmc_sq.txt
