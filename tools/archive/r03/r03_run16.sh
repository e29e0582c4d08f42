#!/bin/bash
# Round-3 GPU call 16: closed-form fit on the stacked merging modes (tests), the closed-form and distributed tests again.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_distributed.py -q -k "normal_eq or merging_modes or stacked" > $O/r3_t_neqmodes.log 2>&1; rc=$?; tail -4 $O/r3_t_neqmodes.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_neqmodes.log | head -40; }
exit $rc
