#!/bin/bash
# Round-3 GPU call 12: the pooled stem chain (bn_act_maxpool): kernel + rewrite tests, the PLeaS / BN-reset tests that run
# through the rewritten sources, the standalone probe, a short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_extras.py tests/test_hip_pipeline.py tests/test_hip_long_horizon.py -q -k "bn_act or pools or reset or train or pleas" > $O/r3_t_pool.log 2>&1; rc=$?; tail -3 $O/r3_t_pool.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_pool.log | head -30; }
timeout -k 10 120 python tools/probe_bn_act.py > $O/r03_probe_bn_act.txt 2>&1; grep -v Warn $O/r03_probe_bn_act.txt | tail -9
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_short.json 2> $O/r03_bench_short.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/r03_bench_short.json')); print(d['value'], d['ms_per_step'], d.get('phases'))"
exit $rc
