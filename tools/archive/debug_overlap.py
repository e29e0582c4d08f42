"""Are the taps of the two-stream source forwards identical to the single-stream ones?  (debug aid)"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import Tiny
from pleas.methods.partial_matching import partial_merge
from pleas.methods.pleas_merging import PleasFitter
t = Tiny("tiny_bottleneck.npz")
m1, m2 = copy.deepcopy(t.m1).cuda(), copy.deepcopy(t.m2).cuda()
perm = t.per_key("am_perm"); costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
data = [x.cuda() for x, _ in t.batches()]
def taps(overlap, fuse=True):
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
    fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, 7, num_classes=10, overlap_sources=overlap, fuse_sources=fuse)
    out = []
    for x in data:
        fit._run_sources(x)
        torch.cuda.synchronize()
        out.append(({k: v.clone() for k, v in fit.tap1.outputs.items()}, {k: v.clone() for k, v in fit.tap2.outputs.items()},
                    {k: v.clone() for k, v in fit.tap1.inputs.items()}, {k: v.clone() for k, v in fit.tap2.inputs.items()}))
        fit.tap1.clear(); fit.tap2.clear()
    fit.finish()
    return out
for fuse in (True, False):
    a, b, c = taps(False, fuse), taps(True, fuse), taps(False, fuse)
    for name, p, q in (("single vs single", a, c), ("single vs two-stream", a, b)):
        worst = 0.0; bad = []
        for i in range(len(p)):
            for j in range(4):
                for k in p[i][j]:
                    d = float((p[i][j][k] - q[i][j][k]).abs().max())
                    if d > 0: bad.append((i, j, k, d))
                    worst = max(worst, d)
        print("fuse=%s %s: worst abs diff %.3e, differing taps %d %s" % (fuse, name, worst, len(bad), bad[:6]))

def fit_weights(overlap, sync=False):
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
    fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, 7, num_classes=10, overlap_sources=overlap)
    for x in data + data:
        fit.step(x)
        if sync: torch.cuda.synchronize()
    return {k: v.clone() for k, v in fit.finish().state_dict().items()}
runs = {"single#1": fit_weights(False), "single#2": fit_weights(False), "two#1": fit_weights(True), "two#2": fit_weights(True),
        "two+sync": fit_weights(True, True), "single+sync": fit_weights(False, True)}
base = runs["single#1"]
for name, w in runs.items():
    bad = [(k, float((w[k].float() - base[k].float()).abs().max())) for k in base if not torch.equal(w[k], base[k])]
    print(name, "differing tensors:", len(bad), bad[:5])
