#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/msr $R/tools/hipbench/mfma_side_rates.hip 2>/dev/null || exit 1
timeout -k 10 60 /tmp/msr > $O/r04_mfma_side_rates.txt 2>&1; cat $O/r04_mfma_side_rates.txt
