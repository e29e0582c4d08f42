#!/bin/bash
# Round-3 GPU call 7: matching with several batches per twin forward: test, probe sweep, short bench at 4 and 8.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py -q -k "matching or api_default" > $O/r3_t_match.log 2>&1; rc=$?; tail -3 $O/r3_t_match.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_match.log | head -30; }
timeout -k 10 300 python tools/probe_matching.py > $O/r03_probe_matching.txt 2>&1; grep "per forward" $O/r03_probe_matching.txt | tail -8
for m in 4 8; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --match-per-forward $m --no-cpu-baseline --no-alt-solver 2> $O/r03_bench_m$m.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('match-per-forward', $m, d['value'], d['phases_s']['matching'], d['checks']['ok'], d['checks']['loss_first_update'])"
  grep "timed region" $O/r03_bench_m$m.err
done
exit $rc
