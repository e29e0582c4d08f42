"""weight_matching (reference pleas/methods/weight_matching.py:59-91) on the MI355X vs the CPU oracle, random-init pairs
(SURVEY.md section 6 probe: ResNet-18 48 LAPs 0.45 s, ResNet-50 666 LAPs 8.6 s on 8 host cores).
Usage: python tools/probe_weight_matching.py [arch ...]  -> gpurun_out/r03_weight_matching.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import pleas_oracle as orc
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.weight_matching import weight_matching

out = {}
for arch in (sys.argv[1:] or ["resnet18", "resnet50", "resnet101"]):
    torch.manual_seed(0)
    m1 = zoo.MODELS[arch](num_classes=1000)
    torch.manual_seed(1)
    m2 = zoo.MODELS[arch](num_classes=1000)
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    sa, sb = m1.state_dict(), m2.state_dict()
    ga, gb = {k: v.cuda() for k, v in sa.items()}, {k: v.cuda() for k, v in sb.items()}
    visits = []
    import builtins
    for rep in range(2):                                  # second run: plans / workspaces warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        perm, costs = weight_matching(spec, ga, gb, max_iter=100, seed=0, verbose=False, return_costs=True)
        torch.cuda.synchronize()
        t_hip = time.perf_counter() - t0
    lines = []
    real_print = builtins.print
    builtins.print = lambda *a, **k: lines.append(a)
    try:
        weight_matching(spec, ga, gb, max_iter=100, seed=0, verbose=True)
    finally:
        builtins.print = real_print
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    operm = orc.weight_matching(spec, sa, sb, max_iter=100, seed=0)
    t_cpu = time.perf_counter() - t0
    operm = operm[0] if isinstance(operm, tuple) else operm
    same = sum(int(torch.equal(perm[k], operm[k])) for k in spec)
    out[arch] = {"groups": len(spec), "lap_visits": len(lines), "hip_s": round(t_hip, 3), "oracle_cpu_s": round(t_cpu, 3),
                 "cpu_threads": torch.get_num_threads(), "groups_with_the_oracles_permutation": "%d / %d" % (same, len(spec))}
    print(arch, out[arch], flush=True)
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(d, exist_ok=True)
json.dump(out, open(os.path.join(d, "r03_weight_matching.json"), "w"), indent=1)
