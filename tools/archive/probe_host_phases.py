"""Host-side set-up costs inside one job (ResNet-101 pair on the device): twin-graph build of the matching phase, fused
source graphs of the PLeaS phase, partial merge, fitter construction -- wall time with the device idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pleas_merging_amd import hip_ops
from pleas_merging_amd.core.compiler import get_permutation_spec
import importlib
am = importlib.import_module("pleas_merging_amd.methods.activation_matching")
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import FrozenSources, PleasFitter

dev = torch.device("cuda")
m1, m2 = bench.build_models("resnet101", dev, 16)
def timed(name, fn, reps=3):
    out = None
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        print("%-34s run %d: %.3f s" % (name, r, time.perf_counter() - t0), flush=True)
    return out
spec = timed("get_permutation_spec", lambda: get_permutation_spec(m1, ((1, 3, 224, 224),)))
arena = am.GroupArena(spec, dev)
timed("build_fused_module (twin graph)", lambda: am.build_fused_module(spec, m1, m2, arena, hip_ops.EPI_NEG_CDIST, True, overlap=True, fuse_bn=True, derive_bn=True))
g = torch.Generator().manual_seed(3)
data = [(torch.randn(16, 3, 224, 224, generator=g).to(dev), None) for _ in range(4)]
perm, costs = am.activation_matching(spec, m1, m2, data, 4, output_costs=True)
timed("FrozenSources (fused source graphs)", lambda: FrozenSources(m1, m2))
m3 = timed("partial_merge", lambda: partial_merge(spec, m1, m2, perm, costs, 0.0, device=dev))
timed("PleasFitter construction", lambda: PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400))
