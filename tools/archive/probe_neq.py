import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import Tiny
from pleas.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.normal_eq import NormalEqFitter, _spd_solve
t = Tiny("tiny_basic.npz")
perm = t.per_key("am_perm"); costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
m1, m2 = copy.deepcopy(t.m1).cuda(), copy.deepcopy(t.m2).cuda()
t0 = time.time(); m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5); print("merge %.2fs" % (time.time() - t0), flush=True)
t0 = time.time(); fit = NormalEqFitter(m1, m2, m3, t.spec, perm, costs, 0.5, 20, num_classes=10); print("init %.2fs" % (time.time() - t0), flush=True)
for i, (x, _) in enumerate(t.batches("xt")[:5]):
    t0 = time.time(); fit.step(x); torch.cuda.synchronize(); print("step %d %.2fs" % (i, time.time() - t0), flush=True)
t0 = time.time(); A = torch.randn(64, 64, device="cuda"); A = A @ A.t() + torch.eye(64, device="cuda"); _spd_solve(A, torch.randn(64, 8, device="cuda"), 1e-6); torch.cuda.synchronize(); print("first cholesky %.2fs" % (time.time() - t0), flush=True)
t0 = time.time(); fit.solve(); torch.cuda.synchronize(); print("solve %.2fs" % (time.time() - t0), flush=True)
