#!/bin/bash
# Round-3 GPU call 25: ResNet-101 over the drivers' full 401 updates, HIP path vs oracle (opt-in test), trajectory kept.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
PLEAS_LONG_RN101=1 timeout -k 10 1100 python -m pytest tests/test_hip_long_horizon.py -q -s -k "rn101" > $O/r3_t_long_rn101.log 2>&1; rc=$?
grep -E "after|passed|failed|Error|assert" $O/r3_t_long_rn101.log | tail -12
exit $rc
