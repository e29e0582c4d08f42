#!/bin/bash
# Round-4 GPU call: rank 0's share of a W-rank job on ONE GPU (collectives skipped): the Adam job and the closed form.
# PROJECTION material, not a multi-GPU measurement.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for W in 2 4 8; do
  timeout -k 10 400 python bench.py --emulate-world $W --steps 2 --warmup 1 --no-cpu-baseline > $O/r04_emulated_rank_w$W.json 2> $O/r04_emulated_rank_w$W.err || { echo "W=$W failed"; tail -5 $O/r04_emulated_rank_w$W.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_emulated_rank_w$W.json").read().strip().splitlines()[-1])
a = d.get("alt_solver_emulated_rank", {})
print("W=$W: Adam job %.3f s per rank (phases %s); closed form: accumulate %s s, solve %s s" % (d["value"], {k: v for k, v in d.get("phases_s", {}).items() if k != "note"}, a.get("accumulate_s"), a.get("solve_s")))
PY
done
