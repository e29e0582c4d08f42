"""Matching phase on ResNet-101, batch 16: steady-state time per batch (host enqueue and wall), measured from the
moments the loop asks for its batches; variants of the twin forward."""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo, hip_ops
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.resnet101().to(dev)
torch.manual_seed(1); m2 = zoo.resnet101().to(dev)
xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(20)]
with torch.no_grad():
    zoo.calibrate_bn(m1, xs[:4]); zoo.calibrate_bn(m2, xs[:4])
m1.eval(); m2.eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
def run(n, **kw):
    stamps = []
    def data():
        for i in range(n + 1):      # the loop stops after n batches; the extra item is never fetched
            stamps.append(time.time())
            yield xs[i % len(xs)], None
    torch.cuda.synchronize(); t0 = time.time()
    accumulate_costs_fused(spec, m1, m2, data(), n, hip_ops.EPI_NEG_CDIST, **kw)
    torch.cuda.synchronize(); t1 = time.time()
    gaps = [b - a for a, b in zip(stamps[10:-1], stamps[11:])]
    return statistics.median(gaps) * 1e3, (t1 - stamps[16]) / (n - 16) * 1e3, (stamps[0] - t0)
for rep in range(3):
    for tag, kw in (("fused, derived, 1 per forward", {"batches_per_forward": 1}), ("fused, derived, 2 per forward", {"batches_per_forward": 2}),
                    ("fused, derived, 4 per forward", {"batches_per_forward": 4}), ("fused, derived, 8 per forward", {"batches_per_forward": 8})):
        host, wall, build = run(64, **kw)
        print("%-27s: host per batch (median) %.2f ms, wall per batch %.2f ms, build %.3f s" % (tag, host, wall, build), flush=True)
