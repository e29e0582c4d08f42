#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { name=$1; shift
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-solver --no-phases "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = {**{d["roofline"]["kernel"][:9]: d["roofline"]}, **{v["kernel"][:9]: v for v in d["roofline_other"].values()}}
print("%-14s job %.3f s (%s) | " % (sys.argv[1], d["value"], d["job_s"]["min"]) + " | ".join("%s %.0f us (%.3f)" % (k, v["avg_launch_us"], v["frac"]) for k, v in sorted(r.items())))
PY
}
for g in 0 2 3 4 6; do run pf$g --sources-per-forward 8 --prefetch-groups $g; done
