#!/bin/bash
# LAP workgroup size A/B through the Python probe: variant libraries built with -DPLEAS_LSAP_THREADS=<T>
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
for T in 256 512 1024; do
  d=/tmp/lap$T; mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -DPLEAS_LSAP_THREADS=$T -c $CS/lsap.hip -o $d/lsap.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v lsap.o) $d/lsap.o
  echo "--- $T threads"; PLEAS_LIB=$d/libpleas_hip.so timeout -k 10 120 python $REPO/tools/probe_lap.py 2>&1 | grep "n=1024\|n=2048\|rn101"
done
