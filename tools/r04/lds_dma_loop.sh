#!/bin/bash
# Round-4 GPU call: tile-loop skeleton with the next chunk loaded straight into LDS (global_load_lds_dwordx4) vs register staging
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/lds_dma_loop lds_dma_loop.hip 2>/dev/null || exit 1
timeout -k 10 120 /tmp/lds_dma_loop > $O/r04_lds_dma_loop.txt 2>&1; rc=$?; cat $O/r04_lds_dma_loop.txt; exit $rc
