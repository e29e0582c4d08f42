import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from conftest import Tiny
from oracle import pleas_oracle as orc
from pleas.methods.partial_matching import partial_merge
from pleas.methods.pleas_merging import train
t = Tiny("tiny_basic.npz")
perm = t.per_key("am_perm"); costs_c = t.per_key("am_cost")
costs = {k: v.cuda() for k, v in costs_c.items()}
for ratio, steps in ((0.0, 1), (0.0, 5), (0.5, 5)):
    m1, m2 = copy.deepcopy(t.m1).cuda(), copy.deepcopy(t.m2).cuda()
    m3 = partial_merge(t.spec, m1, m2, perm, costs, ratio)
    m3 = train(t.batches("xt"), m1, m2, m3, t.spec, perm, costs, ratio, False, steps, None, num_classes=10)
    o3 = orc.partial_merge(t.spec, t.m1, t.m2, perm, costs_c, ratio)
    o0 = {k: v.clone() for k, v in o3.state_dict().items()}
    o3, _ = orc.train(t.batches("xt"), t.m1, t.m2, o3, t.spec, perm, costs_c, ratio, steps, num_classes=10)
    print("ratio", ratio, "steps", steps)
    for k, v in o3.state_dict().items():
        if not v.dtype.is_floating_point or 'running' in k or 'bn' in k or 'downsample.1' in k: continue
        g = m3.state_dict()[k]
        upd = (v - o0[k]).norm()
        err = (g.cpu() - v).norm()
        print("  %-28s |w|=%.3e |upd|=%.3e |err|=%.3e err/upd=%.3e" % (k, v.norm(), upd, err, err / (upd + 1e-30)))
