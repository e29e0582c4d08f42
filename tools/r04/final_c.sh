#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_hip_timed_config.py -x -q -m gpu -s > $O/r04_timed_config_test.log 2>&1; echo "timed-config test rc $?"; tail -2 $O/r04_timed_config_test.log
bash tools/r04/smoke_and_trainmode.sh
bash tools/r04/final_profile.sh c
