#!/bin/bash
# Round-3 GPU call 1: new kernel + parity tests first, then evidence for the closed-form kernels (old vs lag-class form).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -k "normal_eq or cholesky" > $O/r3_t_neq.log 2>&1; rc=$?; tail -3 $O/r3_t_neq.log
[ $rc -ne 0 ] && exit $rc
PLEAS_NEQ_LAG=0 bash tools/run_neq_profile_r03.sh v0_all_blocks > $O/r3_neqprof_v0.log 2>&1; tail -4 $O/r03_neq_v0_all_blocks_plain.txt
bash tools/run_neq_profile_r03.sh v1_lag > $O/r3_neqprof_v1.log 2>&1; tail -4 $O/r03_neq_v1_lag_plain.txt
timeout -k 10 900 python -m pytest tests/test_hip_fullsize_dp.py tests/test_hip_distributed.py tests/test_hip_long_horizon.py -q -s -x > $O/r3_t_dp_long.log 2>&1; rc=$?
grep -E "passed|failed|error|worst|after|one-rank|rank [01]" $O/r3_t_dp_long.log | tail -40
exit $rc
