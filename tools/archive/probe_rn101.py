"""Quick per-phase timing probe on the GPU box (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import activation_matching
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter
from pleas_merging_amd import hip_ops

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda")
def sync(): torch.cuda.synchronize()
t0 = time.time()
torch.manual_seed(0); m1 = zoo.MODELS[arch]().to(dev)
torch.manual_seed(1); m2 = zoo.MODELS[arch]().to(dev)
gen = torch.Generator(device=dev)
def batch(i):
    gen.manual_seed(1000 + i)
    return torch.randn(B, 3, 224, 224, generator=gen, device=dev)
with torch.no_grad():
    zoo.calibrate_bn(m1, [batch(900 + i) for i in range(4)]); zoo.calibrate_bn(m2, [batch(900 + i) for i in range(4)])
sync(); print("models+calibration %.2fs" % (time.time() - t0))
t0 = time.time(); spec = get_permutation_spec(m1, ((1, 3, 224, 224),)); print("spec %.2fs groups=%d" % (time.time() - t0, len(spec)))
data = [(batch(i), None) for i in range(max(nm, ns) + 2)]
sync()
for rep in range(2):
    t0 = time.time(); perm, costs = activation_matching(spec, m1, m2, data, nm, output_costs=True); sync()
    print("activation_matching %d batches: %.3fs (%.1f ms/batch incl. graph build + LAP)" % (nm, time.time() - t0, (time.time() - t0) / nm * 1e3))
t0 = time.time(); outs = hip_ops.solve_lsa_batched(list(costs.values())); sync(); print("LAP batched (%d problems): %.3fs" % (len(outs), time.time() - t0))
t0 = time.time(); m3 = partial_merge(spec, m1, m2, perm, costs, 0.0); sync(); print("partial_merge: %.3fs" % (time.time() - t0))
fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400)
t0 = time.time(); fit.step(data[0][0]); sync(); print("first PLeaS step: %.3fs" % (time.time() - t0))
t0 = time.time()
for i in range(ns): fit.step(data[i + 1][0])
sync(); print("PLeaS steps: %.1f ms/step, loss=%.4e" % ((time.time() - t0) / ns * 1e3, float(fit.loss_now.sum())))
# forward-only reference point
with torch.no_grad():
    t0 = time.time()
    for i in range(ns): m1(data[i][0]); m2(data[i][0])
    sync(); print("two plain forwards: %.1f ms/batch" % ((time.time() - t0) / ns * 1e3))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
