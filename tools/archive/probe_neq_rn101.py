import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo, hip_ops
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.core.utils import make_identity_perm
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.normal_eq import NormalEqFitter
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"; ns = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.MODELS[arch]().to(dev)
torch.manual_seed(1); m2 = zoo.MODELS[arch]().to(dev)
gen = torch.Generator(device=dev)
def batch(i):
    gen.manual_seed(1000 + i); return torch.randn(B, 3, 224, 224, generator=gen, device=dev)
with torch.no_grad():
    zoo.calibrate_bn(m1, [batch(900 + i) for i in range(4)]); zoo.calibrate_bn(m2, [batch(900 + i) for i in range(4)])
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
perm = make_identity_perm(spec); costs = {k: torch.eye(g.size, device=dev) for k, g in spec.items()}
m3 = partial_merge(spec, m1, m2, perm, costs, 0.0)
fit = NormalEqFitter(m1, m2, m3, spec, perm, costs, 0.0, 400)
xs = [batch(i) for i in range(ns + 2)]
fit.step(xs[0]); fit.step(xs[1]); torch.cuda.synchronize()
hip_ops.profile_reset(); hip_ops.profile_enable(True)
t0 = time.time()
for i in range(ns): fit.step(xs[i + 2])
torch.cuda.synchronize(); dt = (time.time() - t0) / ns
hip_ops.profile_enable(False); p = hip_ops.profile_collect()
print("accumulate: %.1f ms/batch" % (dt * 1e3))
for k, v in p.items():
    print("  %-14s %6d launches %8.2f ms/batch %s" % (k, v[0], v[1] / ns, ("%.1f TF/s" % (v[2] / (v[1] * 1e-3) / 1e12)) if v[2] else ""))
t0 = time.time(); fit.solve(); torch.cuda.synchronize(); print("solve: %.2f s" % (time.time() - t0))
print("A arena %.2f GB, B arena %.2f GB, max mem %.1f GB" % (fit.A_flat.numel() * 4 / 2**30, fit.B_flat.numel() * 4 / 2**30, torch.cuda.max_memory_allocated() / 2**30))
