#!/bin/bash
# Short bench runs (1 + 3 jobs, no CPU / library / closed-form legs) on ONE box, one line per run.
# usage: tools/r05/exp_runs.sh <name> <tag> "<ENV=.. assignments or ->" "<bench flags>" [<tag> "<env>" "<flags>" ...]
# output: gpurun_out/r05_exp_<name>.txt (+ the bench lines gpurun_out/r05_exp_<name>_<tag>.json)
set -o pipefail
name=$1; shift
out=gpurun_out/r05_exp_$name.txt
mkdir -p gpurun_out
: > $out
while [ $# -ge 3 ]; do
  tag=$1; envs=$2; flags=$3; shift 3
  [ "$envs" = "-" ] && envs="PLEAS_EXP_TAG=$tag"
  env $envs python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-library-baseline --no-alt-solver $flags \
      > gpurun_out/r05_exp_${name}_$tag.json 2> gpurun_out/r05_exp_${name}_$tag.err || { echo "$tag FAILED" >> $out; tail -5 gpurun_out/r05_exp_${name}_$tag.err; exit 1; }
  python - "$name" "$tag" "$envs $flags" >> $out <<'PY'
import json, sys
name, tag, what = sys.argv[1:4]
d = json.load(open("gpurun_out/r05_exp_%s_%s.json" % (name, tag)))
alt = d.get("alt_arith") or {}
ph = d.get("phases_s") or {}
ak = alt.get("kernels", {})
print("%-16s job %.3f s (min %.3f max %.3f)  alt_arith %s  matching %.3f lap %.3f updates %.3f  sources alone %s  fwd %.0f wgrad %.0f gram %.0f us"
      "  alt: fwd %.0f wgrad %.0f gram %.0f conv2d %.0f us  identical %s  [%s]" % (
    tag, d["value"], d["job_s"]["min"], d["job_s"]["max"], alt.get("job_s"), ph.get("matching", 0), ph.get("lap", 0), ph.get("updates", 0),
    (d.get("vendor") or {}).get("source_forwards_alone_s_per_job"),
    d["roofline_other"].get("conv_fwd", d["roofline"])["avg_launch_us"] if "conv_fwd" in d["roofline_other"] else d["roofline"]["avg_launch_us"],
    d["roofline_other"].get("conv_wgrad", d["roofline"])["avg_launch_us"] if "conv_wgrad" in d["roofline_other"] else d["roofline"]["avg_launch_us"],
    d["roofline_other"]["gram_partial"]["avg_launch_us"],
    ak.get("conv_fwd", {}).get("avg_launch_us", 0), ak.get("conv_wgrad", {}).get("avg_launch_us", 0), ak.get("gram_partial", {}).get("avg_launch_us", 0),
    ak.get("conv2d", {}).get("avg_launch_us", 0), d["checks"].get("last_two_timed_jobs_bit_identical"), what))
PY
done
cat $out
