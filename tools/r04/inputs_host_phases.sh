#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_pipeline.py -x -q -m gpu -k "paired_source or lookahead or replayed or steps_generator or train_adam" > $O/r04_stage_tests.log 2>&1; rc=$?; tail -3 $O/r04_stage_tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_stage_tests.log | head -20; exit $rc; }
for m in resident host; do
  timeout -k 10 400 python bench.py --inputs $m --steps 4 --warmup 1 --no-cpu-baseline --no-alt-solver > $O/r04_inputs_$m.json 2> $O/r04_inputs_$m.err || { tail -3 $O/r04_inputs_$m.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r04_inputs_$m.json"))
print("$m", d["value"], {k:v for k,v in d["phases_s"].items() if k!="note"}, {k:v for k,v in d["inputs"].items() if "s_per_job" in k})
PY
done
