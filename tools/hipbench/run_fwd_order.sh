#!/bin/bash
# fwd_batch_kernel item order experiments (PLEAS_FWD_ORDER): 0 longest first, 1 pseudo-random, 2 long/short folded
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_base fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
for rep in 1 2; do for m in 0 1 2; do echo -n "order=$m: "; PLEAS_FWD_ORDER=$m timeout -k 5 30 /tmp/fwd_base $REPO/tools/hipbench/rn101_layers.txt 20; done; done
