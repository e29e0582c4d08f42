#!/bin/bash
# VGPRs / scratch / occupancy per kernel of one csrc/*.hip file (hipcc remarks), e.g. tools/r05/kernel_resources.sh conv_fwd [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I/root/repo/pleas_merging_amd/csrc "$@" \
  -c /root/repo/pleas_merging_amd/csrc/$f.hip -o /tmp/_res_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "remark: +(Function Name| VGPRs:|VGPRs:|AGPRs:|ScratchSize|Occupancy)" | sed 's/.*remark: *//; s/ \[-Rpass.*//' \
  | paste - - - - - | sed 's/Function Name: //' | c++filt | cut -c1-170
