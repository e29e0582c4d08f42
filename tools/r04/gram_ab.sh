#!/bin/bash
# Round-4 GPU call: matching contraction, before / after the VALU diet of the tile (interior chunks without selects, one
# division per work item, 32-bit byte offsets): kernel tests, then the ResNet-101 replay with both libraries on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O /tmp/before; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gram or cross or cdist" > $O/r04_gram_tests.log 2>&1; rc=$?; tail -3 $O/r04_gram_tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_gram_tests.log | head -30; exit $rc; }
CS=$R/pleas_merging_amd/csrc
cd $CS; for s in *.hip; do src=$s; [ $s = gram.hip ] && src=$R/tools/hipbench/_ab/gram_before.hip; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$CS -c $src -o /tmp/before/${s%.hip}.o 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/before/libpleas_hip.so /tmp/before/*.o || exit 1
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/gram_after gram_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/gram_before gram_batch_rn101.hip -L/tmp/before -lpleas_hip -Wl,-rpath,/tmp/before 2>/dev/null || exit 1
{ for rep in 1 2 3; do for v in before after; do echo -n "$v: "; timeout -k 10 60 /tmp/gram_$v $R/tools/hipbench/rn101_nodes_derived.txt 10 || exit 1; done; done; } > $O/r04_gram_valu_ab.txt 2>&1; cat $O/r04_gram_valu_ab.txt
