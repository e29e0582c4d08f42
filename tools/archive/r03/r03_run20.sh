#!/bin/bash
# Round-3 GPU call 20: the split-bf16 study after pairing the conversions: counters (exact vs split), then the job off / on.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/run_gram_split_pmc_r03.sh > $O/r03_gramsplit_pmc.log 2>&1; grep -E "ms per batch" $O/r03_gramsplit_pmc.log
cd $R
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_exact_short.json 2> $O/r03_bench_exact_short.err; echo "bench rc $?"
PLEAS_GRAM_SPLIT_BF16=1 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_split_study.json 2> $O/r03_bench_split_study.err; echo "bench (study) rc $?"
python -c "
import json
for f in ('r03_bench_exact_short','r03_bench_split_study'):
    d=json.load(open('$O/'+f+'.json')); g=d['roofline'] if 'gram' in d['roofline'].get('kernel','') else d['roofline_other'].get('gram_partial',{})
    print(f, d['value'], d.get('phases_s',{}).get('matching'), g.get('avg_launch_us'), g.get('frac'), d['checks']['ok'], d['metric'][:20])"
