#!/bin/bash
# Round-4 GPU call: kernel + pipeline tests after the wgrad / bias-gradient / H2D changes, then the emulated ranks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_pipeline.py tests/test_hip_extras.py -x -q -m gpu > $O/r04_tests_a.log 2>&1; rc=$?; tail -4 $O/r04_tests_a.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_tests_a.log | head -40; exit $rc; }
bash tools/r04/emulate_world.sh
