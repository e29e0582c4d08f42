"""BASELINE.json configs 2-4 at FULL model size on the MI355X against the CPU oracle.

ResNet-101 (the headline architecture) and ResNet-50 pairs at 224x224 go through the same calls the
benchmark times -- activation matching (all 344 / 174 tracked nodes, the 2048-wide group, the 94-node
residual group, the BatchNorm nodes derived or contracted), the budget sweep {1.0, 1.2, 1.55, 1.8, 2.0}
(``zip_ratios`` -> ``partial_merge``: odd widths 2n-1, frozen gradient-mask blocks, K = 4608 layers) and
PLeaS updates -- and are compared tensor by tensor with ``oracle/pleas_oracle.py`` on the same seeded
inputs.  Batch 2 keeps the oracle at seconds per call (reference: activation_matching.py:139-177,
partial_matching.py:47-202, pleas_merging.py:234-405, experiments/configs/merge_configs.py:25-27).
"""
import copy

import pytest
import torch

from oracle import pleas_oracle as orc
from stem_gate import gate_stem, stem_objective

pytestmark = pytest.mark.gpu

DEGENERATE = "conv1.weight"      # stem: its residual is rounding noise in the reference itself
TOL = 1e-4                       # north-star tolerance: merged weights within 1e-4 rel-fro
BUDGETS = (1.0, 1.2, 1.55, 1.8, 2.0)


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-30))


class Pair:
    """Two calibrated random-init ResNets on the CPU (oracle side), their spec, data and the oracle's matching."""

    def __init__(self, arch: str, batch: int = 2, n_batches: int = 4):
        from pleas_merging_amd import resnet as zoo
        from pleas_merging_amd.core.compiler import get_permutation_spec

        g = torch.Generator().manual_seed(7)
        self.data = [(torch.randn(batch, 3, 224, 224, generator=g), torch.zeros(batch)) for _ in range(n_batches)]
        self.models = []
        for seed in (0, 1):
            torch.manual_seed(seed)
            m = zoo.MODELS[arch](num_classes=1000)
            zoo.calibrate_bn(m, [d[0] for d in self.data])
            self.models.append(m.eval())
        self.m1, self.m2 = self.models
        self.spec = get_permutation_spec(self.m1, ((1, 3, 224, 224),))
        self.want_perm, self.want_costs = orc.activation_matching(self.spec, self.m1, self.m2, self.data, 2,
                                                                  accumulate=True)
        self._gpu = None

    def gpu(self):
        if self._gpu is None:
            self._gpu = (copy.deepcopy(self.m1).cuda(), copy.deepcopy(self.m2).cuda())
        return self._gpu

    def gpu_costs(self):
        return {k: v.cuda() for k, v in self.want_costs.items()}


@pytest.fixture(scope="module")
def rn101():
    return Pair("resnet101")


@pytest.fixture(scope="module")
def rn50():
    return Pair("resnet50")


def _check_matching(p, perm, costs):
    assert len(p.spec) == len(perm) == len(costs)
    flips = {}
    for k in p.spec:
        assert _rel(costs[k], p.want_costs[k]) < TOL, (k, _rel(costs[k], p.want_costs[k]))
        n = int((perm[k] != p.want_perm[k]).sum())
        if n:
            flips[str(k)] = n
    assert not flips, "groups whose assignment differs from the oracle's: %r" % flips


@pytest.mark.parametrize("derive_bn", [True, False])
def test_rn101_activation_matching_vs_oracle(rn101, derive_bn):
    """(a) 71 groups, 344 tracked nodes, 2 accumulated batches: costs within 1e-4 rel-fro and IDENTICAL assignments,
    with the 104 tracked BatchNorm nodes derived from their convolution node (default) and contracted."""
    from pleas_merging_amd import hip_ops
    from pleas_merging_amd.core.solvers import hip_solve_lsa
    from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused, solve_all

    m1, m2 = rn101.gpu()
    assert len(rn101.spec) == 71 and max(g.size for g in rn101.spec.values()) == 2048
    costs = accumulate_costs_fused(rn101.spec, m1, m2, rn101.data, 2, hip_ops.EPI_NEG_CDIST, derive_bn=derive_bn)
    perm = solve_all(costs, hip_solve_lsa)
    _check_matching(rn101, perm, costs)


def test_rn101_api_default_equals_oracle(rn101):
    """The drop-in call itself (defaults: grouped launch, two streams, fused + derived BatchNorm chains)."""
    from pleas.methods.activation_matching import activation_matching

    m1, m2 = rn101.gpu()
    perm, costs = activation_matching(rn101.spec, m1, m2, rn101.data, 2, output_costs=True)
    _check_matching(rn101, perm, costs)
    for k in rn101.spec:
        assert perm[k].dtype == torch.int64 and perm[k].device.type == "cpu" and costs[k].is_cuda
        assert sorted(perm[k].tolist()) == list(range(rn101.spec[k].size))


def _merge_and_train(p, ratios, updates, num_classes=1000):
    """HIP partial_merge + train vs the oracle's, from the ORACLE's perm / costs (so both sides merge the same blocks).
    Returns (worst rel-fro over non-stem float tensors, its key, merged widths)."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    m1, m2 = p.gpu()
    costs = p.gpu_costs()
    m3 = partial_merge(p.spec, m1, m2, p.want_perm, costs, ratios)
    o3 = orc.partial_merge(p.spec, p.m1, p.m2, p.want_perm, p.want_costs, ratios)
    got, want = m3.state_dict(), o3.state_dict()
    assert list(got) == list(want)
    for k in want:
        assert got[k].shape == want[k].shape, k
        if want[k].dtype.is_floating_point:
            assert torch.equal(got[k].cpu(), want[k]), k     # gather / average / halve are exact in fp32
    widths = {k: v.shape[0] for k, v in got.items() if v.dim() == 4}
    init = want[DEGENERATE].clone()
    data = p.data[2:2 + updates]
    m3 = train(data, m1, m2, m3, p.spec, p.want_perm, costs, ratios, False, updates - 1, None, num_classes=num_classes)
    o3, losses = orc.train(data, p.m1, p.m2, o3, p.spec, p.want_perm, p.want_costs, ratios, updates - 1,
                           num_classes=num_classes)
    assert len(losses) == updates
    got, want = m3.state_dict(), o3.state_dict()
    worst, at = 0.0, None
    for k in want:
        if k != DEGENERATE and want[k].dtype.is_floating_point:
            r = _rel(got[k], want[k])
            if r > worst:
                worst, at = r, k
    # the stem's residual is rounding noise in the reference itself: layer objective and travel vs the oracle's
    # (tests/stem_gate.py), not weight for weight
    gate_stem(got[DEGENERATE], init, [want[DEGENERATE]],
              lambda w: stem_objective(p.m1, p.m2, w, p.spec, p.want_perm, p.want_costs, ratios, data, num_classes),
              what="stem")
    return worst, at, widths, got, want


@pytest.mark.parametrize("budget", BUDGETS)
def test_rn101_budget_sweep_partial_merge_and_pleas_vs_oracle(rn101, budget):
    """(b) configs[4]: the five budgets through zip_ratios -> partial_merge -> 2 PLeaS updates.  Merged state dicts are
    bit-equal to the oracle's; every trained tensor (105 layers, K up to 4608 / 9216 at doubled width) < 1e-4 rel-fro."""
    from pleas.methods.extras import zip_ratios

    ratios = zip_ratios(rn101.spec, budget, BUDGETS)
    separate = sum(1 for v in ratios.values() if v == 1.0)
    assert (separate == 0) == (budget == 1.0)
    worst, at, widths, got, want = _merge_and_train(rn101, ratios, 2)
    if budget == 2.0:        # every stage separate: widths 2n-1 (the max-cost unit always stays merged)
        assert widths["layer4.2.conv2.weight"] == 1023 and widths["layer3.0.conv3.weight"] == 2047
    if budget == 1.0:
        assert widths["layer4.2.conv2.weight"] == 512 and got["layer4.0.conv2.weight"].shape[1] * 9 == 4608
    assert worst < TOL, (budget, at, worst)


def test_rn50_full_merge_pleas_vs_oracle(rn50):
    """(c) configs[2]: ResNet-50 pair, budget_ratio 0.0 (full merge), matching + 3 PLeaS updates vs the oracle."""
    from pleas.methods.activation_matching import activation_matching

    m1, m2 = rn50.gpu()
    assert len(rn50.spec) == 37
    perm, costs = activation_matching(rn50.spec, m1, m2, rn50.data, 2, output_costs=True)
    _check_matching(rn50, perm, costs)
    worst, at, widths, _, _ = _merge_and_train(rn50, 0.0, 2)
    assert worst < TOL, (at, worst)
    assert widths["layer4.2.conv3.weight"] == 2048


def test_rn101_mixed_ratio_masks_and_frozen_blocks(rn101):
    """Ratio 0.5 in every group: every layer has merged AND separate units (n_merged < Cout), so the transposed
    gradient-mask blocks (reference pleas_merging.py:57-58) are exercised at full size.  3 updates vs the oracle."""
    worst, at, widths, got, want = _merge_and_train(rn101, 0.5, 3)
    assert widths["layer3.5.conv2.weight"] == 384
    assert worst < TOL, (at, worst)


def test_rn101_planted_permutation_batch16():
    """(d) Size-independent property at the benchmark's batch size: model2 = model1 permuted in all 71 groups (+ small
    noise); matching on 16 x 3 x 224 x 224 batches returns the inverse permutation exactly (20,160 units)."""
    from pleas.core.compiler import get_permutation_spec
    from pleas.core.utils import apply_perm, invert_perm, make_random_perm
    from pleas.methods.activation_matching import activation_matching
    from pleas_merging_amd import resnet as zoo

    torch.manual_seed(0)
    m1 = zoo.MODELS["resnet101"](num_classes=1000).cuda()
    gen = torch.Generator(device="cuda").manual_seed(5)
    data = [(torch.randn(16, 3, 224, 224, device="cuda", generator=gen), None) for _ in range(3)]
    zoo.calibrate_bn(m1, [d[0] for d in data])
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    planted = make_random_perm(spec, torch.Generator().manual_seed(2))
    m2 = copy.deepcopy(m1)
    with torch.no_grad():
        for prm in m2.parameters():
            prm.add_(1e-3 * prm.abs().mean() * torch.randn(prm.shape, device="cuda", generator=gen))
    apply_perm(planted, spec, m2, inplace=True)
    perm = activation_matching(spec, m1, m2, data, 2)
    inv = invert_perm(planted)
    assert sum(g.size for g in spec.values()) == 20160
    for k in spec:
        assert (perm[k] == inv[k]).all(), k
