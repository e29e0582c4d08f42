#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for mode in lap auto; do
  echo "== --gc $mode"
  timeout -k 10 400 python bench.py --steps 12 --warmup 3 --gc $mode --no-cpu-baseline --no-alt-solver --no-phases 2> $O/r03_gc_$mode.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['job_s'], d['checks']['ok'])"
  grep "timed region" $O/r03_gc_$mode.err
done
