import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.core.utils import make_identity_perm
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.resnet101().to(dev).eval()
torch.manual_seed(1); m2 = zoo.resnet101().to(dev).eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
perm = make_identity_perm(spec); costs = {k: torch.eye(g.size, device=dev) for k, g in spec.items()}
m3 = partial_merge(spec, m1, m2, perm, costs, 0.0)
fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400)
x = torch.randn(B, 3, 224, 224, device=dev)
for _ in range(3): fit.step(x)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): fit.step(x)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
