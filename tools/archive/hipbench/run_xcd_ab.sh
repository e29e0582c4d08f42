#!/bin/bash
# A/B of the XCD-aware work-item order (PLEAS_XCD_ORDER=0: plain longest-first) for the grouped forward and
# weight-gradient launches: time (standalone replay) and FETCH_SIZE.
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
for h in fwd_batch_rn101 wgrad_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 0 1; do
  export PLEAS_XCD_ORDER=$mode
  echo -n "xcd_order=$mode fwd:   "; /tmp/fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 20
  echo -n "xcd_order=$mode wgrad: "; /tmp/wgrad_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 20
done
for mode in 0 1; do
  export PLEAS_XCD_ORDER=$mode
  for h in fwd wgrad; do
    rm -rf /tmp/pmcx_$h
    timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcx_$h -o pmc -- /tmp/${h}_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 3 > /tmp/pmcx_$h.log 2>&1 || echo "rocprofv3 failed"
    f=$(find /tmp/pmcx_$h -name "*counter_collection.csv" | head -1)
    echo -n "xcd_order=$mode "; python3 $REPO/tools/pmc_summary.py $f ${h}_batch
  done
done
