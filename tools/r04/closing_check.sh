#!/bin/bash
# Round-4 closing check: smoke(), then the bench exactly as it runs with no flags (traffic stamps must match the sources)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python bench.py > $O/r04_closing_default_bench.json 2> $O/r04_closing_default_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_closing_default_bench.json"))
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "dtype", "scaling", "vs_baseline")})
print("roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "| others", {k: (v["frac"], v["traffic"]) for k, v in d["roofline_other"].items()})
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "| parity ok:", d["checks"]["parity_vs_oracle"]["ok"], "| checks ok:", d["checks"]["ok"])
PY
