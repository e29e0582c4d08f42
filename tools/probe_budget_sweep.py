"""BASELINE.json configs[4]: the ResNet-101 budget sweep {1.0, 1.2, 1.55, 1.8, 2.0} (reference
experiments/configs/merge_configs.py:25-27, run_domainnet.py:34-56): one matching pass, then per budget
zip_ratios -> partial_merge -> 401 PLeaS updates at batch 16, timed on the MI355X (merged widths grow to 2n - 1, so the
merged layers' work grows up to 4x; parity of the same calls at batch 2: tests/test_hip_fullsize.py).
Usage: python tools/probe_budget_sweep.py [updates]  -> gpurun_out/r03_budget_sweep.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import activation_matching
from pleas_merging_amd.methods.extras import zip_ratios
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter

n_updates = int(sys.argv[1]) if len(sys.argv) > 1 else 401
dev = torch.device("cuda")
m1, m2 = bench.build_models("resnet101", dev, 16)
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
pool = bench.Pool(max(100, n_updates), 16, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
perm, costs = activation_matching(spec, m1, m2, pool.loader(0, 100), 100, output_costs=True, batches_per_forward=4)
torch.cuda.synchronize()
out = {"matching_s": round(time.perf_counter() - t0, 3), "updates": n_updates, "budgets": {}}
BUDGETS = (1.0, 1.2, 1.55, 1.8, 2.0)
for budget in BUDGETS:
    ratios = zip_ratios(spec, budget, BUDGETS)
    for rep in range(2):                        # second pass: plans, vendor configurations and allocator pools are warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m3 = partial_merge(spec, m1, m2, perm, costs, ratios, device=dev)
        fit = PleasFitter(m1, m2, m3, spec, perm, costs, ratios, n_updates - 1)
        first = None
        for i in fit.steps([x for x, _ in pool.loader(0, n_updates)], sources_per_forward=8):
            if i == 0:
                first = fit.loss_now.clone()
        last = fit.loss_now.clone()
        m3 = fit.finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    params = sum(p.numel() for n, p in m3.named_parameters())
    fell = int((last[1:] < first[1:]).sum())
    out["budgets"][str(budget)] = {"merge_plus_updates_s": round(dt, 3), "merged_parameters": params,
                                   "separate_groups": sum(1 for v in ratios.values() if v == 1.0),
                                   "layers_whose_loss_fell": "%d / %d" % (fell, first.numel() - 1),
                                   "loss_first": float(first.sum()), "loss_last": float(last.sum())}
    print(budget, out["budgets"][str(budget)], flush=True)
    del fit, m3
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(d, exist_ok=True)
json.dump(out, open(os.path.join(d, "r03_budget_sweep.json"), "w"), indent=1)
