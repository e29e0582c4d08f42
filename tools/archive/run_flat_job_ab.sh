#!/bin/bash
# same-box A/B of the whole job: general forward tile vs flat-shift forms (3 side lanes), default vs 8 hardware queues
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-solver --no-phases > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = {**{d["roofline"]["kernel"][:9]: d["roofline"]}, **{v["kernel"][:9]: v for v in d["roofline_other"].values()}}
print("%-14s job %.3f s | " % (sys.argv[1], d["value"]) + " | ".join("%s %.0f us (%.3f)" % (k, v["avg_launch_us"], v["frac"]) for k, v in sorted(r.items())))
PY
}
run general PLEAS_FWD_FLAT=0
run flat PLEAS_FWD_FLAT=1
run flat_q8 PLEAS_FWD_FLAT=1 GPU_MAX_HW_QUEUES=8
run general_q8 PLEAS_FWD_FLAT=0 GPU_MAX_HW_QUEUES=8
run flat_serial PLEAS_FWD_FLAT=1 PLEAS_FWD_SERIAL=1
