import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.core.utils import make_identity_perm
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter
dev = torch.device("cuda"); B = 16; ns = 20
torch.manual_seed(0); m1 = zoo.resnet101().to(dev).eval()
torch.manual_seed(1); m2 = zoo.resnet101().to(dev).eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
perm = make_identity_perm(spec)
costs = {k: torch.eye(g.size, device=dev) for k, g in spec.items()}
m3 = partial_merge(spec, m1, m2, perm, costs, 0.0)
fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400, fuse_sources='nofuse' not in sys.argv, overlap_sources='nooverlap' not in sys.argv)
print('overlap', 'nooverlap' not in sys.argv, 'fuse_sources', 'nofuse' not in sys.argv, 'graph', 'graph' in sys.argv)
x = torch.randn(B, 3, 224, 224, device=dev)
for _ in range(3): fit.step(x)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(ns): fit.step(x)
torch.cuda.synchronize(); print("PLeaS: %.2f ms/step" % ((time.time() - t0) / ns * 1e3))
xs = [x.clone() for _ in range(ns)]
torch.cuda.synchronize(); t0 = time.time()
for _ in fit.steps(xs, lookahead=True): pass
torch.cuda.synchronize(); print("PLeaS with look-ahead (steps()): %.2f ms/step" % ((time.time() - t0) / ns * 1e3))
# --- is the step CPU-bound?  enqueue time (no sync) vs wall time per step
import time as _t
torch.cuda.synchronize(); t0 = _t.time(); enq = 0.0
for _ in range(ns):
    a = _t.time(); fit.step(x); enq += _t.time() - a
t1 = _t.time(); torch.cuda.synchronize(); t2 = _t.time()
print("enqueue %.2f ms/step, wall %.2f ms/step, GPU drain after last enqueue %.1f ms" % (enq / ns * 1e3, (t2 - t0) / ns * 1e3, (t2 - t1) * 1e3))

# source forwards alone
torch.cuda.synchronize(); t0 = _t.time()
for _ in range(ns): fit._launch_sources(x)
t1 = _t.time(); torch.cuda.synchronize(); t2 = _t.time()
print("sources only: enqueue %.2f ms, wall %.2f ms" % ((t1 - t0) / ns * 1e3, (t2 - t0) / ns * 1e3))
