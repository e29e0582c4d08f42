#!/bin/bash
# A/B of the working-tree conv_fwd.hip against the library already built in csrc/ (build the library BEFORE editing)
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
d=/tmp/fwdvar; mkdir -p $d
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -c $CS/conv_fwd.hip -o $d/conv_fwd.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v conv_fwd.o) $d/conv_fwd.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_base fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_var fwd_batch_rn101.hip -L$d -lpleas_hip -Wl,-rpath,$d 2>/dev/null
for rep in 1 2 3; do
  echo -n "library in csrc/:  "; timeout -k 5 30 /tmp/fwd_base $REPO/tools/hipbench/rn101_layers.txt 20
  echo -n "working-tree file: "; timeout -k 5 30 /tmp/fwd_var $REPO/tools/hipbench/rn101_layers.txt 20
done
