#!/bin/bash
# Round-4 GPU call (STUDY, needs the temporary PLEAS_STUDY_SKIP patch of pleas_merging.py: results are garbage, only the
# clock matters): what does the job save when one of the update's grouped launches is left out?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in none fwd wgrad merge fwd,wgrad,merge none; do
  PLEAS_STUDY_SKIP=$v timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r04_skip_$v.json 2> $O/r04_skip_$v.err
  python - $v <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r04_skip_%s.json" % sys.argv[1]))
    print("skip %-16s: %.3f s per job; updates phase %.3f s" % (sys.argv[1], d["value"], d["phases_s"]["updates"]))
except Exception as e:
    print("skip %s: no line (%s)" % (sys.argv[1], e))
PY
done 2>&1 | tee $O/r04_skip_study.txt
