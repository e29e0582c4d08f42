#!/bin/bash
# fwd_batch_kernel item order experiments (PLEAS_FWD_ORDER): 0 default (channel tile fastest within a layer),
# 3 pixel tile fastest, 1 pseudo-random, 2 long/short folded; with FETCH_SIZE of orders 0 and 3
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_base fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
for rep in 1 2 3; do for m in 0 3; do echo -n "order=$m: "; PLEAS_FWD_ORDER=$m timeout -k 5 30 /tmp/fwd_base $REPO/tools/hipbench/rn101_layers.txt 20; done; done
cd /tmp && export TMPDIR=/tmp
for m in 0 3; do
  rm -rf /tmp/pmco; export PLEAS_FWD_ORDER=$m
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmco -o pmc -- /tmp/fwd_base $REPO/tools/hipbench/rn101_layers.txt 3 > /tmp/pmco.log 2>&1 || echo "rocprofv3 failed"
  f=$(find /tmp/pmco -name "*counter_collection.csv" | head -1); echo -n "order=$m "; python3 $REPO/tools/pmc_summary.py $f fwd_batch
done
