#!/bin/bash
# Round-3 GPU call 21: the other single-GPU configs of BASELINE.json on the closing tree (ResNet-18 / ResNet-50 jobs).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for a in resnet18 resnet50; do
  timeout -k 10 500 python bench.py --arch $a --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_r03_$a.json 2> $O/bench_r03_$a.err || { echo "bench $a failed"; tail -5 $O/bench_r03_$a.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/bench_r03_$a.json')); print('$a', d['value'], d['job_s'], d['phases_s'], d['checks']['ok'])"
done
