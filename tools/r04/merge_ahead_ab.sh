#!/bin/bash
# Round-4 GPU call: the next update's merged inputs written on a stream of their own beside this update's forward
# (PLEAS_MERGE_AHEAD=0: in line, as before): pipeline tests, the timed-configuration parity test, then the job both ways
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
# (the change itself was not kept: no gain; the script needs the patch of that study to mean anything)
timeout -k 10 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_timed_config.py -x -q -m gpu > $O/r04_merge_ahead_tests.log 2>&1; rc=$?; tail -4 $O/r04_merge_ahead_tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_merge_ahead_tests.log | head -40; exit $rc; }
for v in 1 0 1 0; do
  PLEAS_MERGE_AHEAD=$v timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r04_merge_ahead_$v.json 2> $O/r04_merge_ahead_$v.err || { tail -5 $O/r04_merge_ahead_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/r04_merge_ahead_%s.json" % sys.argv[1]))
print("PLEAS_MERGE_AHEAD=%s: %.3f s per job; phases %s; roofline %s" % (sys.argv[1], d["value"], d.get("phases_s"), {k: d["roofline"][k] for k in ("achieved", "frac")}))
PY
done 2>&1 | tee $O/r04_merge_ahead_ab.txt
