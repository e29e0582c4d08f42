#!/bin/bash
# Round-3 GPU call 26: BN-statistics reset with batches sharing a forward (HIP path): tests, then the job's bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_extras.py tests/test_hip_kernels.py tests/test_hip_pipeline.py -q -k "reset or bn_act or bn_fold or source" > $O/r3_t_reset.log 2>&1; rc=$?; tail -3 $O/r3_t_reset.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_reset.log | head -30; exit $rc; }
timeout -k 10 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_reset.json 2> $O/r03_bench_reset.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/r03_bench_reset.json')); print(d['value'], d['bn_reset'])"
