#!/bin/bash
# Round-3 closing GPU call: the whole -m gpu suite, then the round's artefacts with the final code
# (tools/run_final_profile_r03.sh final2: driver bench command, rocprofv3 stats, PMC traffic of the four grouped launches).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r3_t_all.log 2>&1; rc=$?; tail -4 $O/r3_t_all.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_all.log | head -30; }
bash tools/run_final_profile_r03.sh final2 > $O/r03_final2_profile.log 2>&1; tail -12 $O/r03_final2_profile.log
exit $rc
