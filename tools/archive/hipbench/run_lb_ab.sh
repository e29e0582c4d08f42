#!/bin/bash
# A/B: __launch_bounds__(256, 2) on the grouped contraction and weight-gradient kernels (variant sources in /tmp)
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
d=/tmp/lb2; mkdir -p $d
sed 's/__launch_bounds__(kThreads) void gram_batch_kernel/__launch_bounds__(kThreads, 2) void gram_batch_kernel/' $CS/gram.hip > $d/gram.hip
sed 's/__launch_bounds__(cThreads) void wgrad_batch_kernel/__launch_bounds__(cThreads, 2) void wgrad_batch_kernel/' $CS/conv.hip > $d/conv.hip
for f in gram conv; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -c $d/$f.hip -o $d/$f.o 2>/dev/null; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v "gram.o\|conv.o") $d/gram.o $d/conv.o
for h in gram_batch_rn101 wgrad_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/base_$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/lb2_$h $h.hip -L$d -lpleas_hip -Wl,-rpath,$d 2>/dev/null
done
for rep in 1 2; do
  for v in base lb2; do echo -n "$v gram:  "; timeout -k 5 30 /tmp/${v}_gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 10; done
  for v in base lb2; do echo -n "$v wgrad: "; timeout -k 5 30 /tmp/${v}_wgrad_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 20; done
done
