#!/bin/bash
# Round-4 GPU call: updates per source forward beyond 8 (the vendor convolutions are faster per sample at 160 than at 128)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in 8 10 16 10 8 12; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver --sources-per-forward $v > $O/r04_spf_$v.json 2> $O/r04_spf_$v.err || { tail -5 $O/r04_spf_$v.err; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/r04_spf_%s.json" % sys.argv[1]))
print("sources per forward %s: %.3f s per job; updates phase %.3f s; fwd %s" % (sys.argv[1], d["value"], d["phases_s"]["updates"], {k: d["roofline"][k] for k in ("achieved", "frac")}))
PY
done 2>&1 | tee $O/r04_spf_sweep.txt
