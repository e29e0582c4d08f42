"""activation_matching on the ResNet-101 pair, 100 batches of 16 x 3 x 224 x 224, in eval mode and in TRAIN mode (the
reference drivers never call .eval() before matching): fused / derived BatchNorm chains vs module-by-module vendor BatchNorm."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import hip_ops, resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.core.solvers import hip_solve_lsa
from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused, solve_all

dev = torch.device("cuda")
gen = torch.Generator(device=dev)
models = []
for seed in (0, 1):
    torch.manual_seed(seed)
    m = zoo.MODELS["resnet101"](num_classes=1000).to(dev)
    calib = []
    for i in range(4):
        gen.manual_seed(900 + i)
        calib.append(torch.randn(16, 3, 224, 224, generator=gen, device=dev))
    zoo.calibrate_bn(m, calib)
    models.append(m)
spec = get_permutation_spec(models[0], ((1, 3, 224, 224),))
data = []
for b in range(100):
    gen.manual_seed(1000 + b)
    data.append((torch.randn(16, 3, 224, 224, generator=gen, device=dev), None))
results = {}
for name, train, kw in (("eval, fused + derived", False, {}), ("eval, vendor BatchNorm modules", False, {"fuse_bn": False}),
                        ("eval, fused + derived, 10 batches per forward", False, {"batches_per_forward": 10}),
                        ("train, fused + derived, ONE batch per forward", True, {"batches_per_forward": 1}),
                        ("train, fused + derived, 4 batches per forward", True, {"batches_per_forward": 4}),
                        ("train, fused + derived, 10 batches per forward", True, {"batches_per_forward": 10}),
                        ("train, fused + derived (pleas_bn_train_fold)", True, {}),
                        ("train, fused, BatchNorm nodes contracted", True, {"derive_bn": False}),
                        ("train, vendor BatchNorm modules", True, {"fuse_bn": False})):
    ms = [copy.deepcopy(m).train(train) for m in models]
    warm = max(3, 2 * (kw.get("batches_per_forward") or 2))      # two forwards of the timed size: vendor find, plans
    accumulate_costs_fused(spec, ms[0], ms[1], data[:warm], warm, hip_ops.EPI_NEG_CDIST, **kw)
    torch.cuda.synchronize()
    ms = [copy.deepcopy(m).train(train) for m in models]
    t0 = time.perf_counter()
    costs = accumulate_costs_fused(spec, ms[0], ms[1], data, 100, hip_ops.EPI_NEG_CDIST, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    perm = solve_all(costs, hip_solve_lsa)
    results[name] = (dt, perm, costs)
    print("%-48s %.3f s for 100 batches (%.2f ms per batch)" % (name, dt, dt * 10), flush=True)
a, b = results["train, fused + derived (pleas_bn_train_fold)"], results["train, vendor BatchNorm modules"]
flips = sum(int((a[1][k] != b[1][k]).sum()) for k in spec)
rel = max(float((a[2][k] - b[2][k]).norm() / b[2][k].norm()) for k in spec)
print("train mode, fused vs vendor modules: worst cost rel-fro %.2e, differing assignments %d of %d units" % (rel, flips, sum(g.size for g in spec.values())))
a, b = results["train, fused + derived, 10 batches per forward"], results["train, fused + derived, ONE batch per forward"]
flips = sum(int((a[1][k] != b[1][k]).sum()) for k in spec)
rel = max(float((a[2][k] - b[2][k]).norm() / b[2][k].norm()) for k in spec)
print("train mode, 10 batches per forward vs one: worst cost rel-fro %.2e, differing assignments %d of %d units" % (rel, flips, sum(g.size for g in spec.values())))
