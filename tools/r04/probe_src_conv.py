"""Feasibility probe (round 4): the source models' convolutions through the grouped forward kernel (absent targets, dscale 1:
resid = conv(x, w) + bias) against the vendor convolution, layer by layer, ResNet-101 widths, N images per launch."""
import json, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from pleas_merging_amd import hip_ops, resnet as zoo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = zoo.MODELS["resnet101"](num_classes=1000).to(dev).eval()
shapes = {}
hooks = [mod.register_forward_hook(lambda mod, i, o, n=n: shapes.__setitem__(n, (tuple(i[0].shape), tuple(o.shape))))
         for n, mod in m.named_modules() if isinstance(mod, torch.nn.Conv2d)]
with torch.no_grad():
    m(torch.randn(2, 3, 224, 224, device=dev))
for h in hooks:
    h.remove()

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

rows, seen = [], {}
mods = dict(m.named_modules())
tot = {"own": 0.0, "vendor": 0.0, "flop": 0.0}
for name, (si, so) in shapes.items():
    conv = mods[name]
    key = (si[1:], conv.weight.shape, conv.stride, conv.padding)
    if key in seen:
        r = dict(seen[key]); r["name"] = name
    else:
        x = torch.randn((N,) + si[1:], device=dev)
        w = conv.weight.detach()
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Cout, Cin = w.shape[0], w.shape[1]
        kpos = k > 1 and Cin % 32 == 0
        wk = w.permute(0, 2, 3, 1).contiguous() if kpos else w.contiguous()
        y = torch.empty((N,) + so[1:], device=dev)
        none = torch.full((Cout,), -1, dtype=torch.int32, device=dev)
        dummy = torch.zeros((N, 1) + so[2:], device=dev)
        loss = torch.zeros(1, device=dev)
        fb = hip_ops.FwdBatch(dev)
        fb.add(x, wk, None, dummy, dummy, none, none, 0, y, 1.0, 1.0, kernel=(k, k), stride=s, pad=p,
               flags=hip_ops.FwdBatch.KPOS_MAJOR if kpos else 0)
        fb.flush(loss)
        fb.relaunch(loss); fb.relaunch(loss)      # calibration launches of the plan
        want = F.conv2d(x, w, None, s, p)
        err = float((y - want).norm() / want.norm())
        own = timed(lambda: fb.relaunch(loss))
        ven = timed(lambda: F.conv2d(x, w, None, s, p))
        flop = 2.0 * N * so[2] * so[3] * Cout * Cin * k * k
        r = {"name": name, "in": si[1:], "w": tuple(w.shape), "stride": s, "own_ms": round(own, 4), "vendor_ms": round(ven, 4),
             "own_tf": round(flop / own / 1e9, 1), "vendor_tf": round(flop / ven / 1e9, 1), "rel": err, "gflop": flop / 1e9,
             "form": hip_ops.fwd_plan_lanes()["forms"]}
        seen[key] = r
        print(json.dumps(r), flush=True)
    rows.append(r)
    tot["own"] += r["own_ms"]; tot["vendor"] += r["vendor_ms"]; tot["flop"] += r["gflop"]
print(json.dumps({"N": N, "layers": len(rows), "own_ms": round(tot["own"], 3), "vendor_ms": round(tot["vendor"], 3),
                  "gflop": round(tot["flop"], 1), "own_tf": round(tot["flop"] / tot["own"], 1),
                  "vendor_tf": round(tot["flop"] / tot["vendor"], 1), "worst_rel": max(r["rel"] for r in rows)}))
