#!/bin/bash
# Ablation of fwd_batch_kernel on the ResNet-101 layer list (standalone replay): which part of the tile limits it?
# Variant libraries are built with -DPLEAS_FWD_ABLATE=<bits> (see conv_fwd.hip); run on the GPU box.
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
for ab in 0 1 2 4 8 6 14; do
  d=/tmp/ab$ab; mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -DPLEAS_FWD_ABLATE=$ab -c $CS/conv_fwd.hip -o $d/conv_fwd.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v conv_fwd.o) $d/conv_fwd.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o $d/fwd fwd_batch_rn101.hip -L$d -lpleas_hip -Wl,-rpath,$d 2>/dev/null
  echo -n "ablate=$ab: "; timeout -k 10 60 $d/fwd $REPO/tools/hipbench/rn101_layers.txt 10
done
