#!/bin/bash
# Round-5 artefacts (run on the GPU box; outputs under gpurun_out/, copied into profiles/ afterwards):
#   1. the driver's exact bench command                          -> r05_<tag>_bench.json / .err
#   2. rocprofv3 --kernel-trace --stats of a 1 + 3 job run       -> r05_<tag>_bench_kernel_stats.csv
#   3. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of ALL four grouped MFMA launches on their standalone ResNet-101
#      replays (fwd, wgrad, gram with derived BatchNorm nodes, neq) -> r05_<tag>_pmc_<kernel>_<counter>.txt
#   4. SHA-256 of every kernel source as it ran                   -> r05_<tag>_source_sha.json
# Afterwards, in the build container:  python tools/make_traffic_json.py <tag> --round r05 --commit $(git rev-parse --short HEAD)
# Usage: bash tools/r05/final_profile.sh <tag> [skip-bench]
V=${1:-vX}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 - <<PY
import hashlib, json, glob, os
os.chdir("$R")
json.dump({p: hashlib.sha256(open(p, "rb").read()).hexdigest() for p in sorted(glob.glob("pleas_merging_amd/csrc/*.h*"))},
          open("gpurun_out/r05_${V}_source_sha.json", "w"), indent=1)
PY
if [ "$2" != "skip-bench" ]; then
  timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_${V}_bench.json 2> $O/r05_${V}_bench.err
  echo "bench rc $?"; grep "timed region" $O/r05_${V}_bench.err
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof_$V
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$V -o p -- python3 $R/bench.py --steps 3 --warmup 1 \
      --no-cpu-baseline --no-alt-solver --no-phases --no-library-baseline --no-alt-arith > $O/r05_${V}_bench_under_rocprof.json 2> $O/rocprof_$V.err
  cp $(find /tmp/prof_$V -name "*kernel_stats.csv" | head -1) $O/r05_${V}_bench_kernel_stats.csv
  rm -rf /tmp/prof_$V
  grep "timed region" $O/rocprof_$V.err
fi
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
for h in fwd_batch_rn101 wgrad_batch_rn101 gram_batch_rn101 neq_batch_rn101; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/$h $h.hip -L$CS -lpleas_hip -Wl,-rpath,$CS -ldl 2>/dev/null || echo "build of $h failed"
done
cd /tmp && export TMPDIR=/tmp
run() {  # harness, input file, extra arg, tag, kernel filters...
  local h=$1 inp=$2 extra=$3 tag=$4; shift 4
  /tmp/$h $inp 5 $extra | tee $O/r05_${V}_replay_$tag.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcr_${tag}_$c
    timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcr_${tag}_$c -o pmc -- /tmp/$h $inp 3 $extra > /tmp/pmcr_${tag}_$c.log 2>&1 || echo "rocprofv3 $tag $c failed"
    f=$(find /tmp/pmcr_${tag}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f "$@" > $O/r05_${V}_pmc_${tag}_$c.txt
  done
}
# forward / weight gradient: the layer list as the fitter hands it over (strided 1x1 layers as dense layers on the subsampled merge)
run fwd_batch_rn101 $R/tools/hipbench/lists/rn101_dense_downsample.txt "" fwd fwd_batch
run wgrad_batch_rn101 $R/tools/hipbench/lists/rn101_dense_downsample.txt "" wgrad wgrad_batch wgrad_reduce
run gram_batch_rn101 $R/tools/hipbench/rn101_nodes_derived.txt "" gram gram_batch gram_group_reduce
run neq_batch_rn101 $R/tools/hipbench/rn101_layers.txt 0 neq neq_batch neq_reduce
