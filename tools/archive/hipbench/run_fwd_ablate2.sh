#!/bin/bash
# GPU-box half: link each ablated object against the other objects of the library and replay the ResNet-101 layer list
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
for o in _build/conv_fwd_ab*.o; do
  ab=$(basename $o .o | sed 's/conv_fwd_ab//'); d=/tmp/ab$ab; mkdir -p $d
  hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v conv_fwd.o) $o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o $d/fwd fwd_batch_rn101.hip -L$d -lpleas_hip -Wl,-rpath,$d 2>/dev/null
  echo -n "ablate=$ab: "; env "$@" timeout -k 10 60 $d/fwd $REPO/tools/hipbench/rn101_layers.txt 10
done
