#!/bin/bash
# SQ counters (wave cycles, waits, MFMA busy) + GRBM_GUI_ACTIVE (effective clock) of the grouped forward and the grouped
# matching contraction, standalone replays.  One rocprofv3 --pmc pass per kernel, no tracing.
set -e
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_batch_rn101 fwd_batch_rn101.hip -L$REPO/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/gram_batch_rn101 gram_batch_rn101.hip -L$REPO/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$REPO/pleas_merging_amd/csrc
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
rm -rf /tmp/pmcsq_fwd /tmp/pmcsq_gram
timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcsq_fwd -o pmc -- /tmp/fwd_batch_rn101 $REPO/tools/hipbench/rn101_layers.txt 3 > /tmp/pmcsq_fwd.log 2>&1 || { echo "rocprofv3 fwd failed"; tail -5 /tmp/pmcsq_fwd.log; }
f=$(find /tmp/pmcsq_fwd -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f fwd_batch | tee $REPO/gpurun_out/pmc_sq_fwd.txt
timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcsq_gram -o pmc -- /tmp/gram_batch_rn101 $REPO/tools/hipbench/rn101_nodes.txt 3 > /tmp/pmcsq_gram.log 2>&1 || { echo "rocprofv3 gram failed"; tail -5 /tmp/pmcsq_gram.log; }
f=$(find /tmp/pmcsq_gram -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 $REPO/tools/pmc_summary.py $f gram_batch | tee $REPO/gpurun_out/pmc_sq_gram.txt
