#!/bin/bash
# Round-3 GPU call 17: LAP up to n = 4096 (bit-exact vs scipy), LAP timing at 2048 (must not regress) and 4096.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -q -k "lsap or lsa" > $O/r3_t_lap.log 2>&1; rc=$?; tail -3 $O/r3_t_lap.log
[ $rc -ne 0 ] && { grep -E "^E  |Error|FAILED" $O/r3_t_lap.log | head -30; }
timeout -k 10 200 python - > $O/r03_lap_sizes.txt 2>&1 <<'PY'
import time, torch, numpy as np
from pleas_merging_amd import hip_ops
rng = np.random.default_rng(5)
for n in (512, 1024, 2048, 3072, 4096):
    x, y = rng.standard_normal((n, 64)), rng.standard_normal((n, 64))
    c = torch.from_numpy(-np.sqrt(((x[:, None] - y[None]) ** 2).sum(-1)).astype(np.float32)).cuda()
    hip_ops.solve_lsa_batched([c], maximize=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); hip_ops.solve_lsa_batched([c], maximize=True); torch.cuda.synchronize()
    print("n = %4d: %.1f ms" % (n, 1e3 * (time.perf_counter() - t0)), flush=True)
PY
grep -v Warn $O/r03_lap_sizes.txt | tail -6
exit $rc
