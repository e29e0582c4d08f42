#!/bin/bash
# Per-rank critical path of an N-rank job, timed on ONE GPU with the collectives skipped (bench.py --emulate-world).
# Usage (GPU box): bash tools/run_emulate_world.sh "2 4 8"
set -e
mkdir -p gpurun_out
for n in ${1:-2 4 8}; do
  timeout -k 10 300 python bench.py --emulate-world $n --no-cpu-baseline > gpurun_out/emulate_w$n.json 2> gpurun_out/emulate_w$n.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/emulate_w$n.json").read().strip().splitlines()[-1])
print("world $n: %.3f s" % d["value"], {k: (v["launches"], v["total_ms"]) for k, v in d["phases_ms"].items()})
PY
done
