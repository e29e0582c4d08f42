#!/bin/bash
# Round-3 GPU call 11: weight matching with independent visits batched: golden / oracle tests, probe; a short bench for the line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py -q -k "weight_matching or rn18" > $O/r3_t_wm.log 2>&1; rc=$?; tail -3 $O/r3_t_wm.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_wm.log | head -30; }
timeout -k 10 400 python tools/probe_weight_matching.py resnet18 resnet50 resnet101 > $O/r03_weight_matching.log 2>&1; grep -v Warn $O/r03_weight_matching.log | tail -4
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_short.json 2> $O/r03_bench_short.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/r03_bench_short.json')); print(d['value'], d['roofline']['traffic'], d['roofline'].get('hbm_gbps'), {k:(v.get('traffic'), v.get('hbm_gbps'), v.get('traffic_refused')) for k,v in d['roofline_other'].items()})"
timeout -k 10 120 python tools/probe_bn_act.py > $O/r03_probe_bn_act.txt 2>&1; grep -v Warn $O/r03_probe_bn_act.txt | tail -40
exit $rc
