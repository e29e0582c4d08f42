#!/bin/bash
# usage: run_ablate.sh  -> builds variants with -DPLEAS_GRAM_ABLATE=n and runs the big-node shape
cd "$(dirname "$0")"
for v in ${VARIANTS:-0 1 2 3}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../pleas_merging_amd/csrc -DPLEAS_GRAM_ABLATE=$v -o /tmp/gram_ablate_$v gram_ablate.hip 2>/dev/null || { echo build $v failed; exit 1; }
  echo "--- ablate=$v"; /tmp/gram_ablate_$v 1024 196; /tmp/gram_ablate_$v 2048 49; /tmp/gram_ablate_$v 256 3136
done
