#!/bin/bash
# Round-3 GPU call 14: full GPU suite on the tree with the pooled stem pass and the split-bf16 study switch; the study
# probe once more (results kept under profiles/), and the job with the switch off / on.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r3_t_full.log 2>&1; rc=$?; tail -4 $O/r3_t_full.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_full.log | head -30; }
timeout -k 10 300 python tools/probe_gram_split.py --batches 8 --out $O/r03_gram_split.json > $O/r03_gram_split.log 2>&1; echo "probe rc $?"; grep -v Warn $O/r03_gram_split.log | tail -3
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_exact_short.json 2> $O/r03_bench_exact_short.err; echo "bench rc $?"
PLEAS_GRAM_SPLIT_BF16=1 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_split_study.json 2> $O/r03_bench_split_study.err; echo "bench (study) rc $?"
python -c "
import json
for f in ('r03_bench_exact_short','r03_bench_split_study'):
    d=json.load(open('$O/'+f+'.json')); g=d['roofline'] if 'gram' in d['roofline'].get('kernel','') else d['roofline_other'].get('gram_partial',{})
    print(f, d['value'], d.get('phases_s',{}).get('matching'), g.get('avg_launch_us'), g.get('frac'), d['checks']['ok'], d['metric'][:20])"
exit $rc
