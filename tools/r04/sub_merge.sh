#!/bin/bash
# Round-4 GPU call: strided 1x1 layers through the subsampled merge (dense 1x1 layers for the grouped launches): kernel +
# pipeline tests, the timed-configuration parity test, then a short job
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_pipeline.py tests/test_hip_timed_config.py -x -q -m gpu > $O/r04_sub_merge_tests.log 2>&1; rc=$?; tail -4 $O/r04_sub_merge_tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_sub_merge_tests.log | head -40; exit $rc; }
for k in 256 0 256 0; do
PLEAS_FWD_TM64_K=$k timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r04_sub_merge_$k.json 2> $O/r04_sub_merge_$k.err || { tail -5 $O/r04_sub_merge_$k.err; exit 1; }
python - $k <<'PY'
import json, sys
d = json.load(open("gpurun_out/r04_sub_merge_%s.json" % sys.argv[1]))
print("TM64_K=%s: %.3f s per job; updates %.3f s; fwd %s; others %s" % (sys.argv[1], d["value"], d["phases_s"]["updates"], {k: d["roofline"][k] for k in ("achieved", "frac")}, [(o["kernel"], o["frac"]) for o in d.get("roofline_other", [])]))
PY
done 2>&1 | tee $O/r04_sub_merge.txt
