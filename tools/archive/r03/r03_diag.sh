#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-solver --no-phases --updates 41 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['checks']['loss_first_update'], d['checks']['loss_last_update'], d['checks']['layers_whose_loss_fell'], d['kernels_ms'].get('gram_partial'))"; }
run PLEAS_MATCH_PIPELINE=0 PLEAS_FWD_CALIBRATE=0 PLEAS_WGRAD_VECSHIFT=0
run PLEAS_MATCH_PIPELINE=1 PLEAS_FWD_CALIBRATE=0 PLEAS_WGRAD_VECSHIFT=0
run PLEAS_MATCH_PIPELINE=0 PLEAS_FWD_CALIBRATE=1 PLEAS_WGRAD_VECSHIFT=0
run PLEAS_MATCH_PIPELINE=0 PLEAS_FWD_CALIBRATE=0 PLEAS_WGRAD_VECSHIFT=1
