#!/bin/bash
# Round-3 GPU call 10: wgrad with the LDS-staged 16-byte epilogue: kernel tests, replay A/B against the previous library,
# short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_pipeline.py -q -k "wgrad or train or normal_eq or pleas" > $O/r3_t_wgrad.log 2>&1; rc=$?; tail -3 $O/r3_t_wgrad.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_wgrad.log | head -30; }
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/wgrad_batch_rn101 wgrad_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || { echo "build failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do timeout -k 10 60 /tmp/wgrad_batch_rn101 $R/tools/hipbench/rn101_layers.txt 20; done | tee $O/r03_wgrad_epilogue.txt
cd $R
timeout -k 10 500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/r03_bench_short.json 2> $O/r03_bench_short.err; echo "bench rc $?"; grep "timed region\|closed form\|CHECK" $O/r03_bench_short.err
python -c "
import json; d=json.load(open('$O/r03_bench_short.json')); print(d['value'], d['roofline']['avg_launch_us'], d['roofline_other']['conv_wgrad']['avg_launch_us'], d['alt_solver']['accumulate_s'], d['checks']['ok'])"
exit $rc
