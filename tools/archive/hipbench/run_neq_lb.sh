#!/bin/bash
# A/B: __launch_bounds__(256, 2) on neq_batch_kernel, measured through the Python accumulate probe with variant libraries
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
cp $CS/libpleas_hip.so /tmp/libpleas_base.so
d=/tmp/neqlb; mkdir -p $d
sed 's/__launch_bounds__(nThreads) void neq_batch_kernel/__launch_bounds__(nThreads, 2) void neq_batch_kernel/' $CS/normal_eq.hip > $d/normal_eq.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -c $d/normal_eq.hip -o $d/normal_eq.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v normal_eq.o) $d/normal_eq.o
cd $REPO
for v in base lb2 base lb2; do
  if [ $v = lb2 ]; then cp $d/libpleas_hip.so $CS/libpleas_hip.so; else cp /tmp/libpleas_base.so $CS/libpleas_hip.so; fi
  echo -n "$v: "; timeout -k 10 120 python tools/probe_neq_rn101.py resnet101 8 2>&1 | grep "normal_eq\|accumulate" | tr '\n' ' '; echo
done
cp /tmp/libpleas_base.so $CS/libpleas_hip.so
