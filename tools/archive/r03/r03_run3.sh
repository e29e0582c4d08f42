#!/bin/bash
# Round-3 GPU call 3: full -m gpu suite, neq epilogue effect, forward lanes from measured durations (replay A/B + in-job
# timeline), weight-matching probe, one short bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r3_t_all.log 2>&1; rc=$?; tail -4 $O/r3_t_all.log
[ $rc -ne 0 ] && { grep -E "^E |Error|FAILED" $O/r3_t_all.log | head -30; exit $rc; }
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
for h in fwd_batch_rn101 neq_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || { echo "build failed"; exit 1; }
done
L=$R/tools/hipbench/rn101_layers.txt
cd /tmp && export TMPDIR=/tmp
{ for only in 0 1 2 3; do echo "== neq only=$only"; timeout -k 10 60 /tmp/neq_batch_rn101 $L 10 0 $only; done; } > $O/r03_neq_kinds_v3.txt 2>&1; cat $O/r03_neq_kinds_v3.txt
{ for cal in 0 1 0 1; do echo "== PLEAS_FWD_CALIBRATE=$cal"; PLEAS_FWD_CALIBRATE=$cal timeout -k 10 60 /tmp/fwd_batch_rn101 $L 40; done; } > $O/r03_fwd_calibrate_ab.txt 2>&1; cat $O/r03_fwd_calibrate_ab.txt
cd $R
bash tools/prof_forms_in_job.sh > $O/r03_forms_in_job.txt 2>&1; head -16 $O/r03_forms_in_job.txt
timeout -k 10 400 python tools/probe_weight_matching.py resnet18 resnet50 resnet101 > $O/r03_weight_matching.log 2>&1; grep -v Warn $O/r03_weight_matching.log | tail -4
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $O/r03_bench_short.json 2> $O/r03_bench_short.err; echo "bench rc $?"; grep "timed region\|closed form" $O/r03_bench_short.err
