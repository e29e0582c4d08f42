import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import activation_matching, accumulate_costs_fused, solve_all
from pleas_merging_amd.core.solvers import hip_solve_lsa
from pleas_merging_amd.methods.partial_matching import partial_merge, get_blocks, merged_state
from pleas_merging_amd.methods.pleas_merging import PleasFitter
dev = torch.device("cuda"); B = 16
def T():
    torch.cuda.synchronize(); return time.time()
torch.manual_seed(0); m1 = zoo.resnet101().to(dev)
torch.manual_seed(1); m2 = zoo.resnet101().to(dev)
xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(12)]
with torch.no_grad():
    zoo.calibrate_bn(m1, xs[:4]); zoo.calibrate_bn(m2, xs[:4])
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
data = [(x, None) for x in xs]
for rep in range(2):
    t0 = T(); costs = accumulate_costs_fused(spec, m1, m2, data, 10, 1); t1 = T()
    perm = solve_all(costs, hip_solve_lsa); t2 = T()
    blocks = get_blocks(spec, perm, costs, 0.0, False); t3 = T()
    new = merged_state(spec, m1.state_dict(), m2.state_dict(), blocks); t4 = T()
    m3 = partial_merge(spec, m1, m2, perm, costs, 0.0); t5 = T()
    fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400); t6 = T()
    for i in range(5): fit.step(xs[i])
    t7 = T(); fit.finish(); t8 = T()
    print("matching(10 b incl. twin build) %.3f | LAP %.3f | get_blocks %.3f | merged_state %.3f | partial_merge(total) %.3f | fitter init %.3f | 5 steps %.3f | finish %.3f"
          % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7))
