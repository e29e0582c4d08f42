#!/bin/bash
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/gram_batch_rn101 gram_batch_rn101.hip -L$REPO/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$REPO/pleas_merging_amd/csrc 2>/dev/null
for cfg in "112 0" "112 1" "56 1" "224 1" "112 0" "112 1"; do echo "item_chunks xcd = $cfg"; /tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 10 $cfg; done
