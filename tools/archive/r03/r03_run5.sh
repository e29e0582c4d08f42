#!/bin/bash
# Round-3 GPU call 5: kernel tests of the changed tiles, wgrad / neq A/Bs with the under-aligned 16-byte shifted loads,
# forward timeline in the job, short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_pipeline.py -q > $O/r3_t_kern.log 2>&1; rc=$?; tail -3 $O/r3_t_kern.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_kern.log | head -30; }
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
for h in wgrad_batch_rn101 neq_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || { echo "build failed"; exit 1; }
done
L=$R/tools/hipbench/rn101_layers.txt
cd /tmp && export TMPDIR=/tmp
{ for v in 0 1 0 1; do echo "== PLEAS_WGRAD_VECSHIFT=$v"; PLEAS_WGRAD_VECSHIFT=$v timeout -k 10 60 /tmp/wgrad_batch_rn101 $L 20; done; } > $O/r03_wgrad_vecshift_ab.txt 2>&1; cat $O/r03_wgrad_vecshift_ab.txt
{ for only in 0 1; do echo "== neq only=$only"; timeout -k 10 60 /tmp/neq_batch_rn101 $L 10 0 $only; done; } > $O/r03_neq_kinds_v4.txt 2>&1; cat $O/r03_neq_kinds_v4.txt
cd $R
bash tools/prof_forms_in_job.sh > $O/r03_forms_in_job.txt 2>&1; sed -n 3,30p $O/r03_forms_in_job.txt
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $O/r03_bench_short.json 2> $O/r03_bench_short.err; echo "bench rc $?"; grep "timed region\|closed form\|BN reset\|CHECK" $O/r03_bench_short.err
exit $rc
