#!/bin/bash
# gram_batch_kernel item order A/B: 0 longest-first, 1 compact tile block per XCD, 2 whole (node, K range) per XCD
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/gram_batch_rn101 gram_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2 1 2; do echo -n "order=$mode: "; /tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 10 112 $mode; done
for mode in 1 2; do
  rm -rf /tmp/pmcg
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcg -o pmc -- /tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 3 112 $mode > /tmp/pmcg.log 2>&1 || echo "rocprofv3 failed"
  f=$(find /tmp/pmcg -name "*counter_collection.csv" | head -1)
  echo -n "order=$mode "; python3 $REPO/tools/pmc_summary.py $f gram_batch
done
