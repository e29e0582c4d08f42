#!/bin/bash
# Round-3 GPU call 22: k x k flat forward form without the zero-column read (select on the value instead): forward tests,
# standalone replay (was 3.03 ms per update), LDS counters again, short job.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_pipeline.py -q -k "fwd or forward or train or pleas" > $O/r3_t_fwd.log 2>&1; rc=$?; tail -3 $O/r3_t_fwd.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_fwd.log | head -30; exit $rc; }
cd $R/tools/hipbench; CS=$R/pleas_merging_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -o /tmp/fwd_batch_rn101 fwd_batch_rn101.hip -L$CS -lpleas_hip -Wl,-rpath,$CS 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do /tmp/fwd_batch_rn101 $R/tools/hipbench/rn101_layers.txt 20 | tail -1; done | tee $O/r03_fwd_replay_select.txt
C="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES"
rm -rf /tmp/pmcl2; timeout -k 10 60 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcl2 -o pmc -- /tmp/fwd_batch_rn101 $R/tools/hipbench/rn101_layers.txt 3 > /tmp/pmcl2.log 2>&1 || { echo "rocprofv3 failed"; exit 1; }
python3 $R/tools/pmc_summary.py $(find /tmp/pmcl2 -name "*counter_collection.csv" | head -1) fwd_batch > $O/r03_lds_fwd_batch_rn101.txt
grep -E "kernel<6>|kernel<9>" $O/r03_lds_fwd_batch_rn101.txt | grep -E "CONFLICT|IDX_ACTIVE|BUSY" | sed 's/ \+/ /g' | awk '{print $2, $3, $6}'
cd $R
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-solver > $O/r03_bench_select.json 2> $O/r03_bench_select.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/r03_bench_select.json')); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['fwd_forms']['forms'], d['checks'])"
