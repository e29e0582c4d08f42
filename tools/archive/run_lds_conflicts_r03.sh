#!/bin/bash
# LDS-side counters of the grouped forward, weight-gradient and normal-equation launches on their standalone ResNet-101
# replays: bank conflicts vs LDS-active cycles, LDS instructions, MFMA-busy cycles (one --pmc pass per harness).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
for h in fwd_batch_rn101 wgrad_batch_rn101 neq_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || echo "build of $h failed"
done
cd /tmp && export TMPDIR=/tmp
# seven counters: the set that fits one pass on gfx950 (nine aborted the profiler)
C="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES"
for h in fwd_batch_rn101 wgrad_batch_rn101 neq_batch_rn101; do
  extra=""; [ $h = neq_batch_rn101 ] && extra="0"
  rm -rf /tmp/pmcl_$h
  timeout -k 10 60 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcl_$h -o pmc -- /tmp/$h $R/tools/hipbench/rn101_layers.txt 3 $extra > /tmp/pmcl_$h.log 2>&1 || { echo "rocprofv3 $h failed"; tail -3 /tmp/pmcl_$h.log; exit 1; }
  f=$(find /tmp/pmcl_$h -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f fwd_batch wgrad_batch neq_batch > $O/r03_lds_$h.txt
  sed 's/ \+/ /g' $O/r03_lds_$h.txt | awk '{print $1, $2, $3, $5}'
done
