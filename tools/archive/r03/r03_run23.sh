#!/bin/bash
# Round-3 GPU call 23: same-box A/B of the k x k flat forward form: select on the value (tree) vs zero-column read
# (-DPLEAS_FWD_ZEROCOL=1, rebuilt here): standalone replay and short job, A - B - A.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
CS=$R/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_batch_rn101 tools/hipbench/fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
run() {
  for i in 1 2 3; do /tmp/fwd_batch_rn101 $R/tools/hipbench/rn101_layers.txt 20 | tail -1 | sed "s/^/$1 replay: /"; done
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-solver --no-phases > $O/r03_bench_ab_$1.json 2> $O/r03_bench_ab_$1.err || { echo "bench failed"; exit 1; }
  python -c "
import json; d=json.load(open('$O/r03_bench_ab_$1.json')); print('$1 job:', d['value'], d['job_s']['min'], 'fwd us', d['roofline']['avg_launch_us'], d['checks']['loss_last_update'])"
}
rebuild() {  # $1: extra flag
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $1 -I$R/include -I$CS -c $CS/conv_fwd.hip -o $CS/conv_fwd.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libpleas_hip.so $CS/*.o || exit 1
}
ls $CS/*.o > /dev/null 2>&1 || python -m pleas_merging_amd.build
run select1
rebuild -DPLEAS_FWD_ZEROCOL=1
run zerocol
rebuild -DPLEAS_FWD_ZEROCOL=0
run select2
