#!/bin/bash
# Per-rank critical path of an N-rank job, timed on ONE GPU (bench.py --emulate-world): collectives skipped, or replaced
# by a stall of the update stream (ALLREDUCE_US).  Usage (GPU box): bash tools/run_emulate_world.sh "2 4 8" [us] [lookahead] [buckets]
set -e
mkdir -p gpurun_out
US=${2:-0}
LOOK=${3:--1}
BUCKETS=${4:-1}
for n in ${1:-2 4 8}; do
  tag=w${n}_us${US}_la${LOOK}_b${BUCKETS}
  timeout -k 10 300 python bench.py --emulate-world $n --emulate-allreduce-us $US --lookahead $LOOK --no-cpu-baseline \
      > gpurun_out/emulate_$tag.json 2> gpurun_out/emulate_$tag.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/emulate_$tag.json").read().strip().splitlines()[-1])
print("world $n allreduce ${US}us lookahead $LOOK buckets $BUCKETS: %.3f s" % d["value"],
      {k: (v["launches"], v["total_ms"]) for k, v in d["phases_ms"].items() if k in ("conv_fwd", "conv_wgrad")})
PY
done
