#!/bin/bash
# Round-5 GPU call: SQ counters of the weight-gradient / forward replays on ONE class of layers (default: 1x1 layers at 14 x 14,
# one tile form each) under the exact and the split-bf16 arithmetic -- what binds the split loop?  One rocprofv3 --pmc pass per
# counter group and arithmetic, no tracing.   usage: pmc_split.sh <output name> [layer list under tools/hipbench/lists/]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/wgrad_replay wgrad_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_replay fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"
{ for arith in fp32 split_bf16; do for g in 1 2; do
    C=$G1; [ $g = 2 ] && C=$G2
    for k in wgrad fwd; do
      rm -rf /tmp/pmc_$k; export PLEAS_ARITH=$arith
      timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$k -o pmc -- /tmp/${k}_replay $R/tools/hipbench/lists/${2:-rn101_1x1s1_hw196}.txt 3 > /tmp/pmc_$k.log 2>&1 || { echo "rocprofv3 failed"; tail -3 /tmp/pmc_$k.log; exit 1; }
      f=$(find /tmp/pmc_$k -name "*counter_collection.csv" | head -1)
      echo "== $k ${2:-rn101_1x1s1_hw196}, $arith, group $g"; python3 $R/tools/pmc_summary.py $f ${k}_batch
    done
  done; done; } > $O/${1:-r05_pmc_split}.txt 2>&1; tail -80 $O/${1:-r05_pmc_split}.txt
