#!/bin/bash
# Round-4 GPU call: the grouped forward / weight gradient with the three strided 1x1 layers as dense 1x1 layers on the
# subsampled input (what the subsampled merge hands them) vs as strided layers
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc; cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/wgrad_replay wgrad_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS -ldl 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_replay fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS -ldl 2>/dev/null || exit 1
{ for rep in 1 2; do for l in rn101_layers.txt lists/rn101_dense_downsample.txt; do for k in 256 0; do
    echo -n "fwd   $l TM64_K=$k: "; PLEAS_FWD_TM64_K=$k timeout -k 10 60 /tmp/fwd_replay $l 40 || exit 1; done
    echo -n "wgrad $l: "; timeout -k 10 60 /tmp/wgrad_replay $l 20 || exit 1
  done; done; } > $O/r04_dense_ds.txt 2>&1; cat $O/r04_dense_ds.txt
