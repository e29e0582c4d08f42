#!/bin/bash
# Round-3 GPU call 25: ResNet-50 and ResNet-101 over the drivers' full 401 updates, HIP path vs oracle; trajectories kept.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests/test_hip_long_horizon.py -q -s -k "rn101 or rn50" > $O/r3_t_long_rn101.log 2>&1; rc=$?
grep -E "after|passed|failed|Error|assert" $O/r3_t_long_rn101.log | tail -14
exit $rc
