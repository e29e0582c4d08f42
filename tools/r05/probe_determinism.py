"""Round 5 probe: which kernels make the SAME job differ from itself run to run?

(1) one frozen-source forward (FrozenSources.launch) twice on the same batch: every hooked layer's output compared bit for bit,
    in graph order -> the first layer that differs names the vendor solver class;
(2) the same with torch.backends.cudnn.deterministic = True (PyTorch-ROCm hands MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC to
    every convolution descriptor), timed;
(3) the matching twin forward twice -> cost arenas bit for bit.
Usage: python tools/r05/probe_determinism.py [samples=128]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import activation_matching
from pleas_merging_amd.methods.pleas_merging import FrozenSources

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
calib = [torch.randn(16, 3, 224, 224, generator=g).to(dev) for _ in range(2)]
models = []
for seed in (0, 1):
    torch.manual_seed(seed)
    m = zoo.MODELS["resnet101"](num_classes=1000).to(dev)
    zoo.calibrate_bn(m, calib)
    models.append(m.eval())
m1, m2 = models
x = torch.randn(N, 3, 224, 224, generator=g).to(dev)


def forward_twice(tag):
    src = FrozenSources(m1, m2)
    outs = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, (in1, out1), (in2, out2), ev = src.launch(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        outs.append(({k: v.clone() for k, v in out1.items()}, dt))
    src.close()
    diff = []
    for k in outs[1][0]:
        a, b = outs[1][0][k], outs[2][0][k]
        if not torch.equal(a, b):
            mod = dict(m1.named_modules())[k]
            diff.append((k, tuple(a.shape), tuple(mod.weight.shape), float((a - b).abs().max()), float((a.double() - b.double()).norm() / b.double().norm())))
    print(json.dumps({"leg": tag, "samples": N, "forward_s": [round(o[1], 4) for o in outs], "layers": len(outs[1][0]),
                      "layers_differing": len(diff), "first": diff[:6]}), flush=True)


def matching_twice(tag, nb=2, batch=16):
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    gg = torch.Generator().manual_seed(5)
    data = [(torch.randn(batch, 3, 224, 224, generator=gg).to(dev), None) for _ in range(nb + 1)]
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        perm, costs = activation_matching(spec, m1, m2, data, nb, output_costs=True)
        torch.cuda.synchronize()
        res.append((perm, {k: v.clone() for k, v in costs.items()}, time.perf_counter() - t0))
    bad = [str(k) for k in spec if not torch.equal(res[1][1][k], res[2][1][k])]
    flips = [str(k) for k in spec if not torch.equal(res[1][0][k], res[2][0][k])]
    print(json.dumps({"leg": tag, "matching_s": [round(r[2], 3) for r in res], "groups": len(spec), "cost_groups_differing": len(bad),
                      "assignment_groups_differing": len(flips), "first": bad[:5]}), flush=True)


forward_twice("sources, library default")
matching_twice("matching, library default")
torch.backends.cudnn.deterministic = True
forward_twice("sources, cudnn.deterministic")
matching_twice("matching, cudnn.deterministic")
x = x[:16].contiguous()
N = 16
torch.backends.cudnn.deterministic = False
forward_twice("sources 16 samples, default")
torch.backends.cudnn.deterministic = True
forward_twice("sources 16 samples, cudnn.deterministic")
x = x[:4].contiguous()
N = 4
torch.backends.cudnn.deterministic = False
forward_twice("sources 4 samples, default")
matching_twice("matching batch 4, default", 2, 4)
torch.backends.cudnn.deterministic = True
forward_twice("sources 4 samples, cudnn.deterministic")
matching_twice("matching batch 4, cudnn.deterministic", 2, 4)
