#!/bin/bash
# Round-4 GPU call: the streamed forward (conv_fwd_stream.hip) -- kernel tests, then A/B against the one-item-per-workgroup forms
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "fwd" > $O/r04_fwd_stream_tests.log 2>&1; rc=$?; tail -5 $O/r04_fwd_stream_tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_fwd_stream_tests.log | head -30; exit $rc; }
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_rn101 fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || { echo "build failed"; exit 1; }
L=$R/tools/hipbench/rn101_layers.txt
{ for v in 1 0 1 0; do echo -n "PLEAS_FWD_STREAM=$v: "; PLEAS_FWD_STREAM=$v timeout -k 10 60 /tmp/fwd_rn101 $L 20 || exit 1; done; } > $O/r04_fwd_stream_ab.txt 2>&1; cat $O/r04_fwd_stream_ab.txt
