#!/bin/bash
# Round-4 GPU call: source-forward groups enqueued while the LAP kernel runs (2 since round 2), on this round's code
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in 2 3 4 2 3; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver --no-phases --prefetch-groups $v > $O/r04_pf_$v.json 2> $O/r04_pf_$v.err || { tail -5 $O/r04_pf_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/r04_pf_%s.json" % sys.argv[1]))
print("prefetched groups %s: %.3f s per job; fwd %s" % (sys.argv[1], d["value"], d["roofline"]["frac"]))
PY
done 2>&1 | tee $O/r04_prefetch_sweep.txt
