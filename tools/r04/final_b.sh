#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 120 python tools/r04/probe_h2d_overlap.py > $O/r04_h2d_overlap.txt 2>&1; cat $O/r04_h2d_overlap.txt
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-solver --no-phases > $O/r04_bench_nowinograd.json 2> $O/r04_bench_nowinograd.err; grep "timed region" $O/r04_bench_nowinograd.err
bash tools/r04/final_profile.sh b
