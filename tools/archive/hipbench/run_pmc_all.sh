#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, no tracing) of the three grouped MFMA
# kernels on standalone replays of one ResNet-101 batch / update.  Output: gpurun_out/pmc_<kernel>_<counter>.txt
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
for h in fwd_batch_rn101 wgrad_batch_rn101 gram_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
done
/tmp/fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 5
/tmp/wgrad_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 5
/tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 5
/tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes_derived.txt 5
cd /tmp && export TMPDIR=/tmp
run() {  # harness, input file, tag, kernel filters...
  local h=$1 inp=$2 tag=$3; shift 3
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmca_${tag}_$c
    timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmca_${tag}_$c -o pmc -- /tmp/$h $inp 3 > /tmp/pmca_${tag}_$c.log 2>&1 || echo "rocprofv3 $tag $c failed"
    f=$(find /tmp/pmca_${tag}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f "$@" | tee $REPO/gpurun_out/pmc_${tag}_$c.txt
  done
}
run fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt fwd fwd_batch
run wgrad_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt wgrad wgrad_batch wgrad_reduce
run gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt gram gram_batch gram_group_reduce
run gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes_derived.txt gramderived gram_batch gram_group_reduce
