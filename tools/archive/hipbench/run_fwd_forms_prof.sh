#!/bin/bash
# per-form kernel durations of the grouped forward on the ResNet-101 layer list (rocprofv3 kernel trace)
cd "$(dirname "$0")"; REPO=$(cd ../.. && pwd); CS=$REPO/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$REPO/include -o /tmp/fwd_rn101 fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for flat in 0 1; do
  export PLEAS_FWD_FLAT=$flat
  rm -rf /tmp/prof_f$flat
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$flat -o p -- /tmp/fwd_rn101 $REPO/tools/hipbench/rn101_layers.txt 10 > /dev/null 2>&1
  echo "== PLEAS_FWD_FLAT=$flat"
  f=$(find /tmp/prof_f$flat -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "fwd_batch_kernel" in n:
        form = n.split("ILi")[1].split("E")[0] if "ILi" in n else n[:40]
        print("form %s calls %s avg %.1f us total %.2f ms" % (form, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
