#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/mvs $R/tools/hipbench/mfma_valu_share.hip || exit 1
timeout -k 10 60 /tmp/mvs > $O/r04_mfma_valu_share.txt 2>&1; cat $O/r04_mfma_valu_share.txt
