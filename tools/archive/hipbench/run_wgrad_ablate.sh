#!/bin/bash
# What do the scalar, kernel-position-shifted loads of the 3x3 layers cost the weight-gradient kernel?  Variant library:
# every layer takes the unshifted 16-byte path (WRONG results for k x k layers -- timing only).
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
d=/tmp/wgab; mkdir -p $d
sed 's/const bool xs = L.variant \& 4, ys = L.variant \& 8;/const bool xs = L.variant \& 4, ys = false;/' $CS/conv.hip > $d/conv.hip
grep -c "ys = false" $d/conv.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -c $d/conv.hip -o $d/conv.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v "conv.o") $d/conv.o
h=wgrad_batch_rn101
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/base_$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/noshift_$h $h.hip -L$d -lpleas_hip -Wl,-rpath,$d 2>/dev/null
for rep in 1 2; do
  for v in base noshift; do echo -n "$v wgrad: "; timeout -k 5 30 /tmp/${v}_$h $REPO/tools/hipbench/rn101_layers.txt 20; done
done
