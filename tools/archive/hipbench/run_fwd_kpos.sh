#!/bin/bash
# standard vs kernel-position-major weights on the ResNet-101 layer list + per-item cycle stamps of the kpos run
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_batch_rn101 fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
echo -n "standard layout: "; timeout -k 10 60 /tmp/fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 10 0
echo -n "kpos-major:      "; timeout -k 10 60 /tmp/fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 10 1
bash $REPO/tools/hipbench/run_fwd_stamps.sh
