#!/bin/bash
# Round-3 GPU call 6: forward scheduling (low-parallelism forms first, the many-item form levels the lanes): in-job timeline,
# short bench; then the full -m gpu suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/prof_forms_in_job.sh > $O/r03_forms_in_job.txt 2>&1; sed -n 3,32p $O/r03_forms_in_job.txt
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $O/r03_bench_short.json 2> $O/r03_bench_short.err; echo "bench rc $?"; grep "timed region\|closed form\|BN reset\|CHECK" $O/r03_bench_short.err
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r3_t_all.log 2>&1; rc=$?; tail -4 $O/r3_t_all.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_all.log | head -30; }
exit $rc
