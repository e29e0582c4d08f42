#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r04_smoke.log
timeout -k 10 600 python bench.py --match-mode train --steps 3 --warmup 1 --no-alt-solver > $O/r04_bench_trainmode_matching.json 2> $O/r04_bench_trainmode_matching.err; echo "bench rc $?"; grep "timed region" $O/r04_bench_trainmode_matching.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_bench_trainmode_matching.json"))
print(d["value"], d["config"]["matching_mode"], d["phases_s"], d["checks"]["parity_vs_oracle"]["ok"], d["checks"]["ok"])
PY
