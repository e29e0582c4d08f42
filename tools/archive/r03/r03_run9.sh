#!/bin/bash
# Round-3 GPU call 9: the smaller pairs of BASELINE.json's configs (same job) and the ResNet-101 budget sweep (configs[4]).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for a in resnet18 resnet50; do
  timeout -k 10 300 python bench.py --arch $a --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_r03_$a.json 2> $O/bench_r03_$a.err; echo "$a rc $?"; grep "timed region" $O/bench_r03_$a.err
done
timeout -k 10 500 python tools/probe_budget_sweep.py > $O/r03_budget_sweep.log 2>&1; grep -v Warn $O/r03_budget_sweep.log | tail -7
