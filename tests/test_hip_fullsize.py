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

    def __init__(self, arch: str, batch: int = 2, n_batches: int = 5):
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
        self._variant = None

    def variant_matching(self):
        """The oracle against ITSELF: the same restatement with oneDNN convolutions switched off (another summation order
        in every convolution of the 101-layer forwards).  Its disagreement with the default run is the yardstick for what
        any second implementation of the path can be held to at this depth."""
        if self._variant is None:
            with torch.backends.mkldnn.flags(enabled=False):
                self._variant = orc.activation_matching(self.spec, self.m1, self.m2, self.data, 2, accumulate=True)
        return self._variant

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


def test_rn101_train_mode_matching_vs_oracle(rn101):
    """The drivers' real mode (no .eval() before activation_matching): all 104 BatchNorm2d of each ResNet-101 normalise
    with batch statistics.  HIP path (statistics folded per batch, BatchNorm nodes derived) vs the oracle in train mode:
    costs, assignments (near-tie rule of _check_matching) and the running statistics left behind."""
    from pleas.methods.activation_matching import activation_matching

    cpu = [copy.deepcopy(m).train() for m in (rn101.m1, rn101.m2)]
    gpu = [copy.deepcopy(m).cuda() for m in cpu]
    want_perm, want_costs = orc.activation_matching(rn101.spec, cpu[0], cpu[1], rn101.data, 2, accumulate=True)
    perm, costs = activation_matching(rn101.spec, gpu[0], gpu[1], rn101.data, 2, output_costs=True)

    class Want:
        spec, want_perm, want_costs = rn101.spec, None, None

    Want.want_perm, Want.want_costs = want_perm, want_costs
    flips = _check_matching(Want, perm, costs)
    worst = 0.0
    for g, c in zip(gpu, cpu):
        assert g.training
        for (k, a), (_, b) in zip(g.state_dict().items(), c.state_dict().items()):
            if "running_" in k:
                worst = max(worst, _rel(a, b))
            elif k.endswith("num_batches_tracked"):
                assert int(a) == int(b)
    assert worst < 1e-4, worst
    print("train-mode matching: flipped groups", flips, "worst running-stat rel", worst)


def _value(cost, perm):
    return float(cost.double().cpu()[torch.arange(len(perm)), perm].sum())


MAX_FLIPPED_GROUPS = 4      # observed on the MI355X: 2 groups (8 and 9 units, gap <= 1e-7) of 71; + 2 of margin


def _check_matching(p, perm, costs, max_flipped_groups=MAX_FLIPPED_GROUPS):
    """Costs within 1e-4 rel-fro of the oracle's in every group.  Assignments: IDENTICAL to the oracle's, except in
    groups where the optimum is a near-tie that fp32 rounding of the cost matrix decides -- there (i) the HIP assignment
    is exactly what the oracle's LAP (scipy's algorithm) returns on the HIP cost matrix, i.e. the integer path is exact,
    and (ii) under the ORACLE's cost matrix its value is within 1e-6 relative of the oracle's optimum.  The oracle flips
    in the same way against itself (oneDNN off: 9 units of layer4.1.conv2, gap 1e-7; test_oracle_fullsize_spread.py)."""
    assert len(p.spec) == len(perm) == len(costs)
    flips = {}
    for k in p.spec:
        assert _rel(costs[k], p.want_costs[k]) < TOL, (k, _rel(costs[k], p.want_costs[k]))
        assert sorted(perm[k].tolist()) == list(range(p.spec[k].size)), k
        n = int((perm[k] != p.want_perm[k]).sum())
        if n:
            assert (orc.solve_lsa(costs[k].cpu()) == perm[k]).all(), k
            best, mine = _value(p.want_costs[k], p.want_perm[k]), _value(p.want_costs[k], perm[k])
            gap = (best - mine) / abs(best)
            assert 0 <= gap < 1e-6, (k, n, gap)
            flips[str(k)] = (n, gap)
    print("groups with a near-tie assignment flip (units, optimality gap under the oracle's costs):", flips)
    assert len(flips) <= max_flipped_groups, ("flipped groups: %d observed, %d allowed" % (len(flips), max_flipped_groups), flips)
    return flips


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


LR = 5e-4
SHARE = 3e-3     # share of coordinates with a visibly different Adam step: the oracle's own worst tensors reach 1.2e-3


def _oracle_train(p, ratios, data, updates, num_classes, onednn=True):
    with torch.backends.mkldnn.flags(enabled=onednn):
        o3 = orc.partial_merge(p.spec, p.m1, p.m2, p.want_perm, p.want_costs, ratios)
        merged = {k: v.clone() for k, v in o3.state_dict().items()}
        o3, losses = orc.train(data, p.m1, p.m2, o3, p.spec, p.want_perm, p.want_costs, ratios, updates - 1,
                               num_classes=num_classes)
    assert len(losses) == updates
    return merged, {k: v.clone() for k, v in o3.state_dict().items()}


def _merge_and_train(p, ratios, updates, num_classes=1000):
    """HIP partial_merge + train vs the oracle's, from the ORACLE's perm / costs (so both sides merge the same blocks).

    Merged state dicts: bit-equal.  Trained tensors: Adam's first updates are sign-like (m / sqrt(v) = +-1), so a
    coordinate whose gradient is a near-cancellation lands 2 * lr away when rounding flips its sign, and at 101 layers
    the two source forwards of ANY two convolution implementations differ by 1e-5..1e-4 at the top.  The oracle against
    itself (oneDNN off) therefore disagrees by up to 1e-3 rel-fro (fc.weight: 306 of 2M entries flipped; measured in
    test_oracle_fullsize_spread.py).  Gate per tensor:
      * rel-fro(HIP, oracle) <= max(1e-4, 3 x rel-fro(oracle variant, oracle));
      * the share of coordinates that took a visibly different step (|delta| > lr / 10) is at most 3e-3 (the oracle's
        own worst tensors: 1.2e-3) or 3 x the oracle's own share, and all other coordinates together agree to < 1e-4.
    The kernels themselves are held to fp64 on identical inputs in test_rn101_gradients_vs_fp64_on_identical_taps."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    m1, m2 = p.gpu()
    costs = p.gpu_costs()
    data = p.data[2:2 + updates]
    m3 = partial_merge(p.spec, m1, m2, p.want_perm, costs, ratios)
    merged, want = _oracle_train(p, ratios, data, updates, num_classes)
    _, variant = _oracle_train(p, ratios, data, updates, num_classes, onednn=False)
    got = m3.state_dict()
    assert list(got) == list(merged)
    for k in merged:
        assert got[k].shape == merged[k].shape, k
        if merged[k].dtype.is_floating_point:
            assert torch.equal(got[k].cpu(), merged[k]), k     # gather / average / halve are exact in fp32
    widths = {k: v.shape[0] for k, v in got.items() if v.dim() == 4}
    m3 = train(data, m1, m2, m3, p.spec, p.want_perm, costs, ratios, False, updates - 1, None, num_classes=num_classes)
    got = {k: v.cpu() for k, v in m3.state_dict().items()}
    report = {}
    for k in want:
        if k == DEGENERATE or not want[k].dtype.is_floating_point:
            continue
        r, yard = _rel(got[k], want[k]), _rel(variant[k], want[k])
        assert r <= max(TOL, 3 * yard), (k, r, yard)
        stats = []
        for other in (got[k], variant[k]):
            d = (other.double() - want[k].double()).abs()
            affected = d > LR / 10
            stats.append((float(affected.double().mean()), float((d * ~affected).norm() / (want[k].double().norm() + 1e-30))))
        (frac, rest), (yfrac, yrest) = stats
        assert frac <= max(3 * yfrac, SHARE) and rest <= max(TOL, 2 * yrest), (k, stats)
        if r > TOL:
            report[k] = (r, yard, frac, yfrac)
    # the stem's residual is rounding noise in the reference itself: layer objective and travel vs the oracle's
    # (tests/stem_gate.py), not weight for weight
    gate_stem(got[DEGENERATE], merged[DEGENERATE], [want[DEGENERATE], variant[DEGENERATE]],
              lambda w: stem_objective(p.m1, p.m2, w, p.spec, p.want_perm, p.want_costs, ratios, data, num_classes),
              what="stem")
    print("tensors above 1e-4 (rel-fro, oracle's own spread, affected share, oracle's own share):", report)
    return widths, got, want


@pytest.mark.parametrize("budget", BUDGETS)
def test_rn101_budget_sweep_partial_merge_and_pleas_vs_oracle(rn101, budget):
    """(b) configs[4]: the five budgets through zip_ratios -> partial_merge -> 2 PLeaS updates.  Merged state dicts are
    bit-equal to the oracle's; every trained tensor (105 layers, K up to 4608 / 9216 at doubled width) within the gate of _merge_and_train."""
    from pleas.methods.extras import zip_ratios

    ratios = zip_ratios(rn101.spec, budget, BUDGETS)
    separate = sum(1 for v in ratios.values() if v == 1.0)
    assert (separate == 0) == (budget == 1.0)
    widths, got, want = _merge_and_train(rn101, ratios, 2)
    if budget == 2.0:        # every stage separate: widths 2n-1 (the max-cost unit always stays merged)
        assert widths["layer4.2.conv2.weight"] == 1023 and widths["layer3.0.conv3.weight"] == 2047
    if budget == 1.0:
        assert widths["layer4.2.conv2.weight"] == 512 and got["layer4.0.conv2.weight"].shape[1] * 9 == 4608


def test_rn50_full_merge_pleas_vs_oracle(rn50):
    """(c) configs[2]: ResNet-50 pair, budget_ratio 0.0 (full merge), matching + 3 PLeaS updates vs the oracle."""
    from pleas.methods.activation_matching import activation_matching

    m1, m2 = rn50.gpu()
    assert len(rn50.spec) == 37
    perm, costs = activation_matching(rn50.spec, m1, m2, rn50.data, 2, output_costs=True)
    _check_matching(rn50, perm, costs)
    widths, _, _ = _merge_and_train(rn50, 0.0, 2)
    assert widths["layer4.2.conv3.weight"] == 2048


def test_rn18_config1_matching_and_pleas_vs_oracle():
    """configs[1]: ResNet-18 pair (BasicBlock: 3x3 layers only, 12 groups), activation matching + PLeaS at 224 x 224,
    full merge: matching (near-tie rule), merged state dict bit-equal, PLeaS updates by the gate of _merge_and_train."""
    from pleas.methods.activation_matching import activation_matching

    p = Pair("resnet18")
    m1, m2 = p.gpu()
    assert len(p.spec) == 12
    perm, costs = activation_matching(p.spec, m1, m2, p.data, 2, output_costs=True)
    _check_matching(p, perm, costs)
    widths, _, _ = _merge_and_train(p, 0.0, 3)
    assert widths["layer4.1.conv2.weight"] == 512


def test_rn101_mixed_ratio_masks_and_frozen_blocks(rn101):
    """Ratio 0.5 in every group: every layer has merged AND separate units (n_merged < Cout), so the transposed
    gradient-mask blocks (reference pleas_merging.py:57-58) are exercised at full size.  3 updates vs the oracle."""
    widths, got, want = _merge_and_train(rn101, 0.5, 3)
    assert widths["layer3.5.conv2.weight"] == 384


def test_rn101_gradients_vs_fp64_on_identical_taps(rn101):
    """The update's own kernels at ResNet-101 size, isolated from the sources' forward rounding: ONE update of the HIP path
    (grouped merge -> fused MFMA forward + target + residual + loss -> grouped MFMA weight gradient; 105 layers, K up to
    6912, ratio 0.5 so that every layer has merged and separate units) against fp64 autograd of the reference objective
    (pleas_merging.py:281-287) evaluated on the SAME source activations, copied from the GPU taps: rel-fro < 2e-5, or
    within 3x of what fp32 autograd on the CPU reaches against fp64 on that layer."""
    from grad_check import check_update_against_fp64
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    p, ratio = rn101, 0.5
    m1, m2 = p.gpu()
    m3 = partial_merge(p.spec, m1, m2, p.want_perm, p.gpu_costs(), ratio)
    fit = PleasFitter(m1, m2, m3, p.spec, p.want_perm, p.gpu_costs(), ratio, 3, num_classes=1000)
    worst_g, worst_cpu, worst_l, kmax = check_update_against_fp64(fit, m3, p.spec, p.want_perm, p.want_costs, ratio,
                                                                  p.data[2][0], 1000)
    fit.finish()
    assert kmax >= 4608
    print("worst gradient rel-fro vs fp64: HIP %.2e, fp32 CPU autograd %.2e; worst loss rel %.2e, largest K %d"
          % (worst_g, worst_cpu, worst_l, kmax))


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
