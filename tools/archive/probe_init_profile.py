"""cProfile of the one-off host work inside the timed job: twin-graph build, partial_merge, PleasFitter.__init__."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo, hip_ops
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.core.utils import make_identity_perm
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter
import importlib
am = importlib.import_module("pleas_merging_amd.methods.activation_matching")
dev = torch.device("cuda")
torch.manual_seed(0); m1 = zoo.resnet101().to(dev).eval()
torch.manual_seed(1); m2 = zoo.resnet101().to(dev).eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
perm = make_identity_perm(spec); costs = {k: torch.eye(g.size, device=dev) for k, g in spec.items()}
x = torch.randn(16, 3, 224, 224, device=dev)
def job():
    arena = am.GroupArena(spec, dev)
    gm, sinks = am.build_fused_module(spec, m1, m2, arena, hip_ops.EPI_NEG_CDIST, True, overlap=True)
    m3 = partial_merge(spec, m1, m2, perm, costs, 0.0)
    fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400)
    torch.cuda.synchronize()
    return fit
job().finish()
for name, fn in (("twin build", lambda: am.build_fused_module(spec, m1, m2, am.GroupArena(spec, dev), hip_ops.EPI_NEG_CDIST, True, overlap=True)),
                 ("partial_merge", lambda: partial_merge(spec, m1, m2, perm, costs, 0.0))):
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); print("%s: %.3f s" % (name, time.time() - t0))
m3 = partial_merge(spec, m1, m2, perm, costs, 0.0)
torch.cuda.synchronize(); t0 = time.time(); fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400); torch.cuda.synchronize()
print("PleasFitter init: %.3f s" % (time.time() - t0)); fit.finish()
pr = cProfile.Profile(); pr.enable(); f = job(); pr.disable(); f.finish()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
