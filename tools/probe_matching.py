"""Matching phase A/B on ResNet-101, batch 16: vendor BN/add/ReLU modules vs fused tracked BN chains in the twin graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo, hip_ops
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.resnet101().to(dev)
torch.manual_seed(1); m2 = zoo.resnet101().to(dev)
xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(30)]
with torch.no_grad():
    zoo.calibrate_bn(m1, xs[:4]); zoo.calibrate_bn(m2, xs[:4])
m1.eval(); m2.eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
data = [(x, None) for x in xs]
for rep in range(3):
    for fuse in (False, True):
        torch.cuda.synchronize(); t0 = time.time()
        accumulate_costs_fused(spec, m1, m2, data, 30, hip_ops.EPI_NEG_CDIST, fuse_bn=fuse)
        torch.cuda.synchronize(); dt = time.time() - t0
        print("fuse_bn=%s: 30 batches incl. twin build %.3f s" % (fuse, dt), flush=True)
