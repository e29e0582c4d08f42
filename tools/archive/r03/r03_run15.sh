#!/bin/bash
# Round-3 GPU call 15: strided layers through the unfolded merged input (flat 1x1 forms): kernel test, the PLeaS parity
# tests (tiny goldens, rn18 / rn50 vs oracle), then the job with the option off / on.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_pipeline.py tests/test_hip_fullsize.py tests/test_hip_long_horizon.py -q -k "merge or train or pleas or fit or rn18 or rn50 or long" > $O/r3_t_unfold.log 2>&1; rc=$?; tail -4 $O/r3_t_unfold.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_unfold.log | head -30; }
PLEAS_UNFOLD_STRIDED=0 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_unfold_off.json 2> $O/r03_bench_unfold_off.err; echo "bench (off) rc $?"
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_unfold_on.json 2> $O/r03_bench_unfold_on.err; echo "bench (on) rc $?"
python -c "
import json
for f in ('r03_bench_unfold_off','r03_bench_unfold_on'):
    d=json.load(open('$O/'+f+'.json')); k=d['kernels_ms']
    print(f, d['value'], d['phases_s']['updates'], {n:(round(v['total_ms']/v['launches'],3)) for n,v in k.items() if n in ('conv_fwd','conv_wgrad','merge_blocks')}, d['checks']['loss_first_update'], d['checks']['loss_last_update'], d['fwd_forms']['forms'])"
exit $rc
