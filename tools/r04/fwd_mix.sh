#!/bin/bash
# Round-4 GPU call (STUDY): the forward's memory-bound filler form beside the MFMA-bound forms for the whole launch
# (PLEAS_FWD_MIX bit 0: filler = one unit on its own lane, launched first; bit 1: 128-row forms ask for 82 KB of LDS, i.e. one
# per CU; bit 2: the filler's lane at the lowest stream priority)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc; cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_replay fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS -ldl 2>/dev/null || exit 1
{ for m in 0 1 5 3 7 2 0; do echo -n "PLEAS_FWD_MIX=$m: "; PLEAS_FWD_MIX=$m timeout -k 10 60 /tmp/fwd_replay rn101_layers.txt 40 || exit 1; done; } > $O/r04_fwd_mix.txt 2>&1; cat $O/r04_fwd_mix.txt
