#!/bin/bash
# Round-4 GPU call: the new kernel test, the timed-configuration parity test, a short bench with the parity leg
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "fwd" > $O/r04_fwd_tests.log 2>&1; rc=$?; tail -3 $O/r04_fwd_tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_fwd_tests.log | head -30; exit $rc; }
timeout -k 10 500 python -m pytest tests/test_hip_timed_config.py -x -q -m gpu -s > $O/r04_timed_config_test.log 2>&1; rc=$?; tail -3 $O/r04_timed_config_test.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r04_timed_config_test.log | head -30; }
timeout -k 10 900 python bench.py --steps 3 --warmup 1 > $O/r04_bench_short.json 2> $O/r04_bench_short.err; echo "bench rc $?"; grep "timed region\|closed form\|BN reset\|CHECK\|cpu" $O/r04_bench_short.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_bench_short.json"))
print(json.dumps({k:d[k] for k in ("value","roofline","cpu_baseline")}, indent=1)[:1500]); print(json.dumps(d["checks"], indent=1)[:3000])
PY
