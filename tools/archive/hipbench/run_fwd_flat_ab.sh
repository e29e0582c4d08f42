#!/bin/bash
# A/B of the flat-shift tile (PLEAS_FWD_FLAT=1, default) against the general tile (PLEAS_FWD_FLAT=0) on the ResNet-101
# layer list, same library; optional extra env in "$@" (e.g. PLEAS_XCD_ORDER=1)
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_rn101 fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
for rep in 1 2 3; do
  echo -n "general tile:    "; env PLEAS_FWD_FLAT=0 "$@" timeout -k 5 60 /tmp/fwd_rn101 $REPO/tools/hipbench/rn101_layers.txt 20
  echo -n "flat-shift tile: "; env PLEAS_FWD_FLAT=1 "$@" timeout -k 5 60 /tmp/fwd_rn101 $REPO/tools/hipbench/rn101_layers.txt 20
done
