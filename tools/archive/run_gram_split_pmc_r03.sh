#!/bin/bash
# STUDY evidence: SQ / LDS / memory-pipe counters of the matching contraction on its standalone ResNet-101
# replay (tools/hipbench/gram_batch_rn101.hip), exact fp32-MFMA tile vs the split-bf16 tile (PLEAS_GRAM_SPLIT_BF16=1).
# Outputs: gpurun_out/r03_gramsplit_<exact|split>_{plain,pmc_sq,pmc_lds,pmc_mem}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/gram_rn101 gram_batch_rn101.hip -L$R/pleas_merging_amd/csrc -lpleas_hip -Wl,-rpath,$R/pleas_merging_amd/csrc 2>/dev/null || { echo "build failed"; exit 1; }
N=$R/tools/hipbench/rn101_nodes_derived.txt
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
LDS="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES"
MEM="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum"
for mode in exact split; do
  if [ $mode = split ]; then export PLEAS_GRAM_SPLIT_BF16=1; else export PLEAS_GRAM_SPLIT_BF16=0; fi
  timeout -k 10 120 /tmp/gram_rn101 $N 5 | tee $O/r03_gramsplit_${mode}_plain.txt || exit 1
  for set in sq lds mem; do
    case $set in sq) C=$SQ;; lds) C=$LDS;; mem) C=$MEM;; esac
    rm -rf /tmp/pmcg_${mode}_$set
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcg_${mode}_$set -o pmc -- /tmp/gram_rn101 $N 3 > /tmp/pmcg_${mode}_$set.log 2>&1 || { echo "rocprofv3 $mode $set failed"; tail -5 /tmp/pmcg_${mode}_$set.log; continue; }
    f=$(find /tmp/pmcg_${mode}_$set -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f gram_batch > $O/r03_gramsplit_${mode}_pmc_$set.txt
    cat $O/r03_gramsplit_${mode}_pmc_$set.txt
  done
done
