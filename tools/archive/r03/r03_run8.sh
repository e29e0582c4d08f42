#!/bin/bash
# Round-3 GPU call 8: knobs re-checked with this round's kernels (short benches, same box): look-ahead, prefetch groups,
# matching batches per forward.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { echo "== $*"; timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver --no-phases "$@" 2>$O/r03_knob.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['job_s']['min'], d['job_s']['max'], d['checks']['ok'], d['roofline']['avg_launch_us'], d['roofline_other']['conv_wgrad']['avg_launch_us'])"; }
run
run --lookahead 1
run --prefetch-groups 3
run --match-per-forward 2
run --sources-per-forward 4
run
