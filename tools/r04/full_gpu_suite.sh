#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $O/r04_full_gpu_suite.log 2>&1; rc=$?; tail -6 $O/r04_full_gpu_suite.log
[ $rc -ne 0 ] && grep -E "^E  |Error|FAILED" $O/r04_full_gpu_suite.log | head -40
exit $rc
