#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { name=$1; shift
  timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-solver --no-phases "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  grep "warm-up job 1\|timed region" gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s job %.3f s (min %s) vendor sources alone %.3f ms per update" % (sys.argv[1], d["value"], d["job_s"]["min"], d["vendor"]["ms_per_update"]))
PY
}
run find0 --miopen-find 0
run find1 --miopen-find 1
