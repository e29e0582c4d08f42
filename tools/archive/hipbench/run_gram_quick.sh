#!/bin/bash
# matching contraction replay: all nodes contracted vs BatchNorm nodes derived (current library)
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/gram_batch_rn101 gram_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
for rep in 1 2; do
  /tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 10
  /tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes_derived.txt 10
done
