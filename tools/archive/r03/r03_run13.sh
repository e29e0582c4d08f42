#!/bin/bash
# Round-3 GPU call 13: the split-bf16 study of the matching contraction: switch test, then the probe.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -q -k "gram" > $O/r3_t_split.log 2>&1; rc=$?; tail -3 $O/r3_t_split.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_split.log | head -30; }
timeout -k 10 700 python tools/probe_gram_split.py --batches 8 --out $O/r03_gram_split.json > $O/r03_gram_split.log 2>&1; echo "probe rc $?"; grep -v Warn $O/r03_gram_split.log | tail -24
exit $rc
