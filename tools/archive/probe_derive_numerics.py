"""ResNet-101 pair, batch 16: matching costs with BatchNorm nodes derived vs contracted (relative difference per group,
and whether the assignments agree)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo, hip_ops
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.core.solvers import hip_solve_lsa
from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused, solve_all
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.resnet101().to(dev)
torch.manual_seed(1); m2 = zoo.resnet101().to(dev)
xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(8)]
with torch.no_grad():
    zoo.calibrate_bn(m1, xs[:4]); zoo.calibrate_bn(m2, xs[:4])
m1.eval(); m2.eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
data = [(x, None) for x in xs]
a = {k: v.clone() for k, v in accumulate_costs_fused(spec, m1, m2, data, 8, hip_ops.EPI_NEG_CDIST, derive_bn=False).items()}
b = {k: v.clone() for k, v in accumulate_costs_fused(spec, m1, m2, data, 8, hip_ops.EPI_NEG_CDIST, derive_bn=True).items()}
c = {k: v.clone() for k, v in accumulate_costs_fused(spec, m1, m2, data, 8, hip_ops.EPI_NEG_CDIST, fuse_bn=False).items()}
rel = lambda p, q: float((p.double() - q.double()).norm() / q.double().norm())
worst = max(rel(b[k], a[k]) for k in spec); worst_vendor = max(rel(a[k], c[k]) for k in spec)
pa, pb, pc = solve_all(a, hip_solve_lsa), solve_all(b, hip_solve_lsa), solve_all(c, hip_solve_lsa)
same_ab = sum(int((pa[k] == pb[k]).all()) for k in spec); same_ac = sum(int((pa[k] == pc[k]).all()) for k in spec)
mism = sum(int((pa[k] != pb[k]).sum()) for k in spec); tot = sum(pa[k].numel() for k in spec)
print("derived vs contracted: worst rel-fro %.2e; folded-BN vs vendor-BN modules: %.2e" % (worst, worst_vendor))
print("assignments equal in %d / %d groups (derived vs contracted), %d / %d (folded vs vendor); differing units %d of %d" %
      (same_ab, len(spec), same_ac, len(spec), mism, tot))
