#!/bin/bash
# Round-4 GPU call: when and where every work item of the grouped forward / weight gradient runs (study build of the library
# with PLEAS_FWD_TIMELINE / PLEAS_WGRAD_TIMELINE: wall-clock stamps per item) -- what is the ~0.3 ms per launch that no class
# of layers accounts for (profiles/r04_marginal_class.txt)?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O /tmp/tl; CS=$R/pleas_merging_amd/csrc; cd $CS
for s in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$CS -DPLEAS_FWD_TIMELINE=1 -DPLEAS_WGRAD_TIMELINE=1 -c $s -o /tmp/tl/${s%.hip}.o 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/tl/libpleas_hip.so /tmp/tl/*.o || exit 1
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/tl/fwd fwd_batch_rn101.hip -L/tmp/tl -lpleas_hip -Wl,-rpath,/tmp/tl -ldl 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/tl/wgrad wgrad_batch_rn101.hip -L/tmp/tl -lpleas_hip -Wl,-rpath,/tmp/tl -ldl 2>/dev/null || exit 1
PLEAS_TIMELINE_OUT=$O/r04_timeline_fwd.bin timeout -k 10 60 /tmp/tl/fwd rn101_layers.txt 30 || exit 1
PLEAS_TIMELINE_OUT=$O/r04_timeline_wgrad.bin timeout -k 10 60 /tmp/tl/wgrad rn101_layers.txt 20 || exit 1
