"""The yardstick of the full-size parity tests: how far the CPU oracle (the restatement pinned by the reference's own
fixtures) lands from ITSELF on a ResNet-101 pair when nothing changes but the summation order inside its convolutions
(oneDNN on / off).  No GPU involved.

Findings this test pins (they are why tests/test_hip_fullsize.py gates the way it does):
  * matching costs agree to < 1e-4 rel-fro, yet assignments can differ in a group whose optimum is a near-tie
    (relative optimality gap < 1e-6 under either cost matrix);
  * after Adam's sign-like first update, trained tensors differ by up to ~1e-3 rel-fro (fc.weight) -- a few hundred of
    2M coordinates per tensor whose gradient is a near-cancellation (|g| within rounding of 0, or of Adam's eps) take
    a different step, up to 2 * lr apart -- while all other coordinates together agree to < 1e-4 rel-fro.  "Within 1e-4 rel-fro of the reference" is therefore attainable on the shallow golden
    fixtures (where it is enforced, test_hip_pipeline.py) but not between ANY two arithmetic variants of the reference
    path at ResNet-101 depth, batch 2.
"""
import torch

from oracle import pleas_oracle as orc


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_oracle_against_itself_resnet101():
    from pleas_merging_amd import resnet as zoo
    from pleas_merging_amd.core.compiler import get_permutation_spec

    g = torch.Generator().manual_seed(7)
    data = [(torch.randn(2, 3, 224, 224, generator=g), None) for _ in range(4)]
    models = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = zoo.MODELS["resnet101"](num_classes=1000)
        zoo.calibrate_bn(m, [d[0] for d in data])
        models.append(m.eval())
    m1, m2 = models
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    runs = []
    for onednn in (True, False):
        with torch.backends.mkldnn.flags(enabled=onednn):
            perm, costs = orc.activation_matching(spec, m1, m2, data, 2, accumulate=True)
            runs.append((perm, costs))
    (p0, c0), (p1, c1) = runs
    worst_cost = max(_rel(c1[k], c0[k]) for k in spec)
    assert 1e-6 < worst_cost < 1e-4, worst_cost          # 4e-5 at layer4: the forwards drift apart with depth
    flipped = {}
    for k in spec:
        n = int((p0[k] != p1[k]).sum())
        if n:
            idx = torch.arange(len(p0[k]))
            best, other = float(c0[k].double()[idx, p0[k]].sum()), float(c0[k].double()[idx, p1[k]].sum())
            flipped[str(k)] = (n, (best - other) / abs(best))
    assert all(0 <= gap < 1e-6 for _, gap in flipped.values()), flipped       # near-ties only
    trained = []
    for onednn in (True, False):
        with torch.backends.mkldnn.flags(enabled=onednn):
            o3 = orc.partial_merge(spec, m1, m2, p0, c0, 0.0)
            o3, _ = orc.train(data[2:4], m1, m2, o3, spec, p0, c0, 0.0, 1)
            trained.append({k: v.clone() for k, v in o3.state_dict().items()})
    a, b = trained
    spread = {k: _rel(a[k], b[k]) for k in a if a[k].dtype.is_floating_point and k != "conv1.weight"}
    assert max(spread.values()) > 1e-4, max(spread.values())     # the north-star tolerance is out of reach at this depth
    assert max(spread.values()) < 5e-3
    for k, r in spread.items():
        d = (a[k].double() - b[k].double()).abs()
        affected = d > 5e-5       # lr / 10: the coordinate took a visibly different step (opposite sign: 2 * lr)
        assert float(affected.double().mean()) < 3e-3, k
        assert float((d * ~affected).norm() / (b[k].double().norm() + 1e-30)) < 1e-4, k   # the rest: north-star tolerance
    print("assignment flips:", flipped, "| worst tensors:", sorted(spread.items(), key=lambda kv: -kv[1])[:3])
