#!/bin/bash
# Round-4 GPU call: which short-K 1x1 layers take 64-row tiles (PLEAS_FWD_TM64_K: Kd up to which they do; default 256)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc; cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_replay fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS -ldl 2>/dev/null || exit 1
{ for k in 256 0 64 128 512 1024 256; do echo -n "PLEAS_FWD_TM64_K=$k: "; PLEAS_FWD_TM64_K=$k timeout -k 10 60 /tmp/fwd_replay rn101_layers.txt 40 || exit 1; done; } > $O/r04_fwd_tm64.txt 2>&1; cat $O/r04_fwd_tm64.txt
