#!/bin/bash
# Round-3 GPU call 19: matching batches per twin forward 4 (default) vs 5 vs 10 on the same box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for m in 10 20 25 10; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver --match-per-forward $m > $O/r03_mpf_$m.json 2> $O/r03_mpf_$m.err || { echo "bench $m failed"; tail -5 $O/r03_mpf_$m.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/r03_mpf_$m.json')); print('per forward $m:', d['value'], d['phases_s']['matching'], d['checks']['ok'])"
done
