#!/bin/bash
# Round-3 GPU call 2: neq per-kind timings with / without the XCD-aware order, then the new parity tests (no -x).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -k "normal_eq or cholesky" > $O/r3_t_neq.log 2>&1; rc=$?; tail -3 $O/r3_t_neq.log
[ $rc -ne 0 ] && exit $rc
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/neq_rn101 neq_batch_rn101.hip -L$R/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$R/pleas_merging_amd/csrc 2>/dev/null || { echo "build failed"; exit 1; }
L=$R/tools/hipbench/rn101_layers.txt
cd /tmp && export TMPDIR=/tmp
{
for x in 1 0; do for lag in 1 0; do for only in 0 1 2 3; do
  echo "== PLEAS_XCD_ORDER=$x PLEAS_NEQ_LAG=$lag only=$only"
  PLEAS_XCD_ORDER=$x PLEAS_NEQ_LAG=$lag timeout -k 10 60 /tmp/neq_rn101 $L 10 0 $only
done; done; done
} > $O/r03_neq_kinds.txt 2>&1
cat $O/r03_neq_kinds.txt
for c in FETCH_SIZE; do
  rm -rf /tmp/pmcn_$c
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcn_$c -o pmc -- /tmp/neq_rn101 $L 3 0 > /tmp/pmcn_$c.log 2>&1 || { echo "rocprofv3 $c failed"; tail -5 /tmp/pmcn_$c.log; }
  f=$(find /tmp/pmcn_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f neq_ | tee $O/r03_neq_v2_xcd_pmc_$c.txt
done
cd $R
timeout -k 10 1000 python -m pytest tests/test_hip_fullsize_dp.py tests/test_hip_distributed.py tests/test_hip_long_horizon.py tests/test_hip_extras.py -q -s > $O/r3_t_dp_long.log 2>&1; rc=$?
grep -E "passed|failed|error|worst|after|one-rank|rank [01]|twice|BN reset|^FAILED|^ERROR" $O/r3_t_dp_long.log | tail -60
exit $rc
