#!/bin/bash
# Round-4 GPU call: per-role cycle accounting of the streamed forward (stamped build of the library in /tmp on the box)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O /tmp/stamped; CS=$R/pleas_merging_amd/csrc
cd $CS
for s in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$CS -DPLEAS_FWDS_STAMPS=1 -c $s -o /tmp/stamped/${s%.hip}.o & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/stamped/libpleas_hip.so /tmp/stamped/*.o || exit 1
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_stamps fwd_stream_stamps.hip -L/tmp/stamped -lpleas_hip -Wl,-rpath,/tmp/stamped || exit 1
timeout -k 10 120 /tmp/fwd_stamps $R/tools/hipbench/rn101_layers.txt 10 > $O/r04_fwd_stream_stamps.txt 2>&1; cat $O/r04_fwd_stream_stamps.txt
