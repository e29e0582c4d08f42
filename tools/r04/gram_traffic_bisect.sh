#!/bin/bash
# Round-4 GPU call: is the matching contraction's extra HBM traffic since round 1 (8.9 -> 13.4 GB per batch) the
# desynchronisation of a (node, K range)'s tiles by commit 4ffa37f (tiles that store no row norms skip that arithmetic and
# run ahead of the ones that do)?  Same replay, the library built with every tile doing the norm arithmetic again.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O /tmp/norms; CS=$R/pleas_merging_amd/csrc
cd $CS; for s in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$CS -DPLEAS_GRAM_NORMS_ALWAYS=1 -c $s -o /tmp/norms/${s%.hip}.o 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/norms/libpleas_hip.so /tmp/norms/*.o || exit 1
cd $R/tools/hipbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/gram_now gram_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/gram_norms gram_batch_rn101.hip -L/tmp/norms -lpleas_hip -Wl,-rpath,/tmp/norms 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
L=$R/tools/hipbench/rn101_nodes_derived.txt
out=$O/r04_gram_traffic_bisect.txt; : > $out
for v in now norms now norms; do
  echo "== $v (norms = every tile does the row-norm arithmetic)" >> $out
  timeout -k 10 60 /tmp/gram_$v $L 10 >> $out 2>&1 || exit 1
  rm -rf /tmp/pmc_f
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o pmc -- /tmp/gram_$v $L 3 > /tmp/pmc_f.log 2>&1 || echo "rocprofv3 failed" >> $out
  f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f gram_batch >> $out
done
cat $out
