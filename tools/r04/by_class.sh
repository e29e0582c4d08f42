#!/bin/bash
# Round-4 GPU call: the grouped weight gradient and forward on SUBSETS of the ResNet-101 layer list (1x1 stride 1 / 3x3 stride 1 /
# the rest; by image size): which class of layers holds the launch below the matching contraction's 0.77?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; CS=$R/pleas_merging_amd/csrc; cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/wgrad_replay wgrad_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_replay fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
{ for c in 1x1s1 1x1s1_big 1x1s1_hw196 1x1s1_hw49 3x3s1 3x3s1_big 3x3s1_14 rest; do
    echo -n "wgrad $c: "; timeout -k 10 60 /tmp/wgrad_replay lists/rn101_$c.txt 20 || exit 1
    echo -n "fwd   $c: "; timeout -k 10 60 /tmp/fwd_replay lists/rn101_$c.txt 30 || exit 1
  done
  echo -n "wgrad all: "; timeout -k 10 60 /tmp/wgrad_replay rn101_layers.txt 20
  echo -n "fwd   all: "; timeout -k 10 60 /tmp/fwd_replay rn101_layers.txt 30; } > $O/r04_by_class.txt 2>&1; cat $O/r04_by_class.txt
