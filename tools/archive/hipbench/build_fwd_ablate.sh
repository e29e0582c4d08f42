#!/bin/bash
# Build-container half of the forward ablation: cross-compile conv_fwd.hip with -DPLEAS_FWD_ABLATE=<bits> into
# tools/hipbench/_build/conv_fwd_ab<bits>.o (objects travel with gpurun; linking happens on the GPU box, run_fwd_ablate2.sh)
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
mkdir -p _build
for ab in ${@:-0 1 2 4 8 6 14}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$CS -DPLEAS_FWD_ABLATE=$ab -c $CS/conv_fwd.hip -o _build/conv_fwd_ab$ab.o 2>/dev/null &
done
wait
ls -la _build
