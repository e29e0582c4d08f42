import sys, copy, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import Tiny
from pleas.methods.activation_matching import activation_matching
t = Tiny("tiny_bottleneck.npz")
m1, m2 = copy.deepcopy(t.m1).cuda().train(), copy.deepcopy(t.m2).cuda().train()
perm, costs = activation_matching(t.spec, m1, m2, t.batches(), 3, output_costs=True)
print("train-mode matching ok:", m1.training, m2.training, all(torch.isfinite(c).all().item() for c in costs.values()))
m1.eval()
perm, costs = activation_matching(t.spec, m1, m2, t.batches(), 3, output_costs=True)
print("mixed-mode matching ok:", m1.training, m2.training, all(torch.isfinite(c).all().item() for c in costs.values()))
