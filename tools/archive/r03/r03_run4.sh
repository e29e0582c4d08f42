#!/bin/bash
# Round-3 GPU call 4: full suite after the pruning / plan caches / pipelined matching / wgrad vector shift / forward units,
# then A/Bs: matching pipeline, wgrad shifted loader, forward timeline in the job, short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r3_t_all.log 2>&1; rc=$?; tail -4 $O/r3_t_all.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_all.log | head -30; }
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
for h in wgrad_batch_rn101 fwd_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || { echo "build failed"; exit 1; }
done
L=$R/tools/hipbench/rn101_layers.txt
cd /tmp && export TMPDIR=/tmp
{ for v in 0 1 0 1; do echo "== PLEAS_WGRAD_VECSHIFT=$v"; PLEAS_WGRAD_VECSHIFT=$v timeout -k 10 60 /tmp/wgrad_batch_rn101 $L 20; done; } > $O/r03_wgrad_vecshift_ab.txt 2>&1; cat $O/r03_wgrad_vecshift_ab.txt
{ for cal in 0 1; do echo "== PLEAS_FWD_CALIBRATE=$cal"; PLEAS_FWD_CALIBRATE=$cal timeout -k 10 60 /tmp/fwd_batch_rn101 $L 40; done; } > $O/r03_fwd_calibrate_ab.txt 2>&1; cat $O/r03_fwd_calibrate_ab.txt
cd $R
timeout -k 10 300 python tools/probe_matching.py > $O/r03_probe_matching.txt 2>&1; grep -v Warn $O/r03_probe_matching.txt | tail -8
bash tools/prof_forms_in_job.sh > $O/r03_forms_in_job.txt 2>&1; sed -n 3,24p $O/r03_forms_in_job.txt
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $O/r03_bench_short.json 2> $O/r03_bench_short.err; echo "bench rc $?"; grep "timed region\|closed form\|BN reset" $O/r03_bench_short.err
exit $rc
