#!/bin/bash
# Round-3 GPU call 24: train-mode matching with several batches per twin forward (ONE statistics launch and ONE pass per
# BatchNorm for all batches of the forward): tests, then the ResNet-101 probe (eval / train, 1 / 4 / 10 batches per forward).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py tests/test_hip_kernels.py tests/test_hip_extras.py -q -k "matching or train_mode or bn_act or bn_train or bn_fold or derived or reset" > $O/r3_t_trainmode.log 2>&1; rc=$?; tail -3 $O/r3_t_trainmode.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_trainmode.log | head -30; exit $rc; }
timeout -k 10 500 python tools/probe_trainmode_matching.py > $O/r03_trainmode_matching.txt 2>&1; echo "probe rc $?"; grep -v Warn $O/r03_trainmode_matching.txt | tail -12
