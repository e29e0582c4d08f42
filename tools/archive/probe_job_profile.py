"""cProfile of a shortened job (20 matching batches + 41 updates, ResNet-101, batch 16): host time by function."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import activation_matching
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter, prepare_sources
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.resnet101().to(dev)
torch.manual_seed(1); m2 = zoo.resnet101().to(dev)
xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(41)]
with torch.no_grad():
    zoo.calibrate_bn(m1, xs[:4]); zoo.calibrate_bn(m2, xs[:4])
m1.eval(); m2.eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
def job(nm, nu):
    early = {}
    perm, costs = activation_matching(spec, m1, m2, [(x, None) for x in xs[:nm]], nm, output_costs=True,
                                      while_solving=lambda: early.update(s=prepare_sources(m1, m2)))
    m3 = partial_merge(spec, m1, m2, perm, costs, 0.0, device=dev)
    fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, nu - 1, fused_sources=early["s"])
    for _ in fit.steps(xs[:nu]): pass
    return fit.finish()
job(3, 3); torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.time(); pr.enable(); job(20, 41); torch.cuda.synchronize(); pr.disable()
print("job(20 matching batches, 41 updates): %.3f s" % (time.time() - t0))
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
