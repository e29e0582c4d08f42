#!/bin/bash
# Round-4 GPU call: the source models' convolutions through the grouped forward kernel vs the vendor's, per layer
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 500 python tools/r04/probe_src_conv.py 128 > $O/r04_probe_src_conv_128.txt 2>&1 || { tail -20 $O/r04_probe_src_conv_128.txt; exit 1; }
tail -1 $O/r04_probe_src_conv_128.txt
timeout -k 10 300 python tools/r04/probe_src_conv.py 160 > $O/r04_probe_src_conv_160.txt 2>&1 || { tail -20 $O/r04_probe_src_conv_160.txt; exit 1; }
tail -1 $O/r04_probe_src_conv_160.txt
