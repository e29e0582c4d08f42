#!/bin/bash
# Round-4 GPU call: where do the matching contraction's 13.4 GB per batch (2.6 x algorithmic) come from?  Time and
# FETCH_SIZE / WRITE_SIZE of the ResNet-101 replay for several K ranges per work item and XCD orders.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/gram_rn101 gram_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
L=$R/tools/hipbench/rn101_nodes_derived.txt
out=$O/r04_gram_traffic_sweep.txt; : > $out
for cfg in "112 2" "112 0" "112 1" "56 2" "28 2" "16 2" "28 0"; do
  set -- $cfg
  echo "== item_chunks=$1 xcd_order=$2" >> $out
  timeout -k 10 60 /tmp/gram_rn101 $L 10 $1 $2 >> $out 2>&1 || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- /tmp/gram_rn101 $L 3 $1 $2 > /tmp/pmc_$c.log 2>&1 || echo "rocprofv3 failed" >> $out
    f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f gram_batch gram_group_reduce >> $out
  done
done
cat $out
