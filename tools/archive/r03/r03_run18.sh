#!/bin/bash
# Round-3 GPU call 18: the whole GPU suite and the smoke entry on the closing tree.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/r3_t_full.log 2>&1; rc=$?; tail -4 $O/r3_t_full.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_full.log | head -30; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r3_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r3_smoke.log
exit $rc
