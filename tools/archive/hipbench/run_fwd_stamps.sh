#!/bin/bash
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
d=/tmp/ab16; mkdir -p $d
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -DPLEAS_FWD_ABLATE=16 -c $CS/conv_fwd.hip -o $d/conv_fwd.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libpleas_hip.so $(ls $CS/*.o | grep -v conv_fwd.o) $d/conv_fwd.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o $d/fwd_stamps fwd_stamps.hip -L$d -lpleas_hip -Wl,-rpath,$d 2>/dev/null
timeout -k 10 60 $d/fwd_stamps $REPO/tools/hipbench/rn101_layers.txt
