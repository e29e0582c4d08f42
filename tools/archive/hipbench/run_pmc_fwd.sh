#!/bin/bash
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_batch_rn101 fwd_batch_rn101.hip -L$REPO/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$REPO/pleas_merging_amd/csrc
/tmp/fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 5
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcf_$c
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcf_$c -o pmc -- /tmp/fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 3 > /tmp/pmcf_$c.log 2>&1 || echo "rocprofv3 $c failed"
  f=$(find /tmp/pmcf_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f fwd_batch | tee $REPO/gpurun_out/pmc_fwd_$c.txt
done
