#!/bin/bash
# does the grouped forward slow down when its targets are windows of 8x larger tap tensors (sources_per_forward = 8)?
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_rn101 fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
for cfg in "1 0" "2 1" "8 0" "8 3" "8 7"; do set -- $cfg
  echo -n "tap group $1 window $2: "; PLEAS_TAP_GROUP=$1 PLEAS_TAP_WINDOW=$2 timeout -k 5 120 /tmp/fwd_rn101 $REPO/tools/hipbench/rn101_layers.txt 20
done
