#!/bin/bash
# Round-5 GPU call: the grouped launches on the ResNet-101 replay (batch 16) with the exact fp32 MFMA arithmetic and under
# PLEAS_ARITH=split_bf16, same box, plus the weight gradient / forward by class of layers.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc; cd $R/tools/hipbench
for k in wgrad fwd gram; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/${k}_replay ${k}_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
done
{ for arith in fp32 split_bf16; do
    echo -n "wgrad all   $arith: "; PLEAS_ARITH=$arith timeout -k 10 60 /tmp/wgrad_replay rn101_layers.txt 20 || exit 1
    echo -n "fwd   all   $arith: "; PLEAS_ARITH=$arith timeout -k 10 60 /tmp/fwd_replay rn101_layers.txt 30 || exit 1
    echo -n "gram  batch $arith: "; PLEAS_ARITH=$arith timeout -k 10 60 /tmp/gram_replay rn101_nodes_derived.txt 10 || exit 1
    for c in 1x1s1_big 1x1s1_hw196 3x3s1_big 3x3s1_14; do
      echo -n "wgrad $c $arith: "; PLEAS_ARITH=$arith timeout -k 10 60 /tmp/wgrad_replay lists/rn101_$c.txt 20 || exit 1
      echo -n "fwd   $c $arith: "; PLEAS_ARITH=$arith timeout -k 10 60 /tmp/fwd_replay lists/rn101_$c.txt 30 || exit 1
    done
  done; } > $O/${1:-r05_split_ab}.txt 2>&1; cat $O/${1:-r05_split_ab}.txt
