#!/bin/bash
# Builds the standalone replay and collects FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes.
set -e
cd "$(dirname "$0")"
REPO=$(cd ../.. && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/gram_batch_rn101 gram_batch_rn101.hip -L$REPO/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$REPO/pleas_merging_amd/csrc
/tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 5
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- /tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 3 > /tmp/pmc_$c.log 2>&1 || echo "rocprofv3 $c failed"
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f gram_batch gram_group_reduce | tee $REPO/gpurun_out/pmc_gram_$c.txt
done
