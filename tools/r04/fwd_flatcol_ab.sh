#!/bin/bash
# Round-4 GPU call: grouped forward (default forms), before / after the flat image as [column][36] with a zero row (KIND 2):
# tiles, division-free tap offsets, 32-bit byte offsets in the epilogue): kernel tests, then the ResNet-101 replay, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O /tmp/before; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "fwd" > $O/r04_fwd_flatcol_tests.log 2>&1; rc=$?; tail -3 $O/r04_fwd_flatcol_tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_fwd_flatcol_tests.log | head -30; exit $rc; }
CS=$R/pleas_merging_amd/csrc
cd $CS; for s in *.hip; do src=$s; [ $s = conv_fwd.hip ] && src=$R/tools/hipbench/_ab/conv_fwd_before.hip; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$CS -c $src -o /tmp/before/${s%.hip}.o 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/before/libpleas_hip.so /tmp/before/*.o || exit 1
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_after fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_before fwd_batch_rn101.hip -L/tmp/before -lpleas_hip -Wl,-rpath,/tmp/before 2>/dev/null || exit 1
L=$R/tools/hipbench/rn101_layers.txt
{ for rep in 1 2 3; do for v in before after; do echo -n "$v: "; timeout -k 10 60 /tmp/fwd_$v $L 30 || exit 1; done; done; } > $O/r04_fwd_flatcol_ab.txt 2>&1; cat $O/r04_fwd_flatcol_ab.txt
