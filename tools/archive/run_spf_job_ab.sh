#!/bin/bash
# same-box A/B of the whole job: updates per source forward 2 vs 8 (and the general tile for reference)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-solver --no-phases $ARGS > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = {**{d["roofline"]["kernel"][:9]: d["roofline"]}, **{v["kernel"][:9]: v for v in d["roofline_other"].values()}}
print("%-14s job %.3f s | " % (sys.argv[1], d["value"]) + " | ".join("%s %.0f us (%.3f)" % (k, v["avg_launch_us"], v["frac"]) for k, v in sorted(r.items())))
PY
}
ARGS="--sources-per-forward 2 --prefetch-groups 24" run spf2 A=1
ARGS="--sources-per-forward 8 --prefetch-groups 6" run spf8 A=1
ARGS="--sources-per-forward 8 --prefetch-groups 6" run spf8_general PLEAS_FWD_FLAT=0
ARGS="--sources-per-forward 8 --prefetch-groups 0" run spf8_noprefetch A=1
