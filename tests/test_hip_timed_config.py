"""The configuration ``bench.py`` TIMES, under the oracle at full size (missing 1).

ResNet-101 pair, batch 16 x 3 x 224 x 224, the bench's knobs: ten matching batches through ONE twin forward
(``batches_per_forward=10``, 160 samples) -> batched LAP with the frozen sources' first group prefetched from
``while_solving`` (``FrozenSources.prefetch``, exactly ``bench.run_job``) -> full merge -> eight PLeaS updates whose
batches go through the sources as ONE forward (``sources_per_forward=8``, 128 samples), cosine schedule of the 401-update
job.  Compared with ``oracle/pleas_oracle.py`` on the same batches (reference: activation_matching.py:119-134,
pleas_merging.py:265-291, :367-375): group costs, assignments (near-tie rule of test_hip_fullsize._check_matching),
merged state dict bit-equal, trained tensors by the gate of test_hip_fullsize._merge_and_train.
"""
import copy
import json
import os

import pytest
import torch

from conftest import REPO, _usable_cores
from oracle import pleas_oracle as orc
from stem_gate import gate_stem, stem_objective
import test_hip_fullsize as fs

pytestmark = pytest.mark.gpu

BATCH, N_MATCH, N_UPDATES = 16, 10, 8
T_MAX = 400                      # CosineAnnealingLR(T_max=MAX_STEPS) of the 401-update job (pleas_merging.py:358)


class _Want:
    pass


ANCHOR_FACTOR = 1.5      # rel-fro(HIP, fp64) <= 1.5 x rel-fro(fp32 oracle, fp64)


def fp64_gate(costs_hip, costs_ref, costs64, got, want, variant, want64):
    """Per group cost and per trained tensor: distance of the HIP path to the fp64 anchor against the distance of the
    reference's own fp32 arithmetic (the oracle; for trained tensors the larger of its two oneDNN variants) to the same anchor.

    * costs are sums of rounding errors: per group hip <= ANCHOR_FACTOR x ref (floor 2e-6: the kernel tests' floor for cross features
      against fp64 -- a group at 1.0e-6 where the oracle sits at 6.4e-7 is fp32 rounding, not a finding), no exceptions;
    * trained tensors carry Adam's sign-like first steps -- a coordinate whose gradient is a near-cancellation lands 2 lr
      away in ANY fp32 run, and which coordinates those are is a draw per run -- so per tensor the ratio scatters around 1:
      the MEDIAN ratio must be <= 1.25 (as accurate as the reference on the whole), at most one tensor in 20 above
      ANCHOR_FACTOR x its own reference distance (floor 1e-4), none above 3 x the model's largest reference distance
      (measured, two model draws: median 0.90 / 0.96; one tensor of 105 above 1.5 x each time -- layer3.8.conv2.weight at 1.5 x,
      layer3.11.conv2.weight at 2.15 x its own and 1.8 x the model's largest, which itself moved 2 x between the draws)."""
    rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30))
    c = {str(k): (rel(costs_hip[k], costs64[k]), rel(costs_ref[k], costs64[k])) for k in costs64}
    t = {k: (rel(got[k], want64[k]), max(rel(want[k], want64[k]), rel(variant[k], want64[k]))) for k in got}
    ratios = sorted(a / max(b, 1e-30) for a, b in t.values())
    ref_max = max(b for _, b in t.values())
    over_c = {k: v for k, v in c.items() if v[0] > max(2e-6, ANCHOR_FACTOR * v[1])}
    over_t = {k: v for k, v in t.items() if v[0] > max(1e-4, ANCHOR_FACTOR * v[1])}
    summary = {"groups": len(c), "worst_cost_hip_vs_fp64": max(a for a, _ in c.values()),
               "worst_cost_oracle_fp32_vs_fp64": max(b for _, b in c.values()),
               "cost_groups_above_factor": over_c, "tensors": len(t),
               "worst_tensor_hip_vs_fp64": max(a for a, _ in t.values()), "worst_tensor_oracle_fp32_vs_fp64": ref_max,
               "median_ratio_hip_over_oracle": ratios[len(ratios) // 2], "max_ratio": ratios[-1],
               "tensors_above_1e-4_vs_fp64": {"hip": sum(1 for a, _ in t.values() if a > 1e-4),
                                               "oracle_fp32": sum(1 for _, b in t.values() if b > 1e-4)},
               "tensors_above_factor": over_t, "factor": ANCHOR_FACTOR}
    ok = (not over_c and ratios[len(ratios) // 2] <= 1.25 and len(over_t) <= max(1, len(t) // 20)
          and all(a <= max(1e-4, 3 * ref_max) for a, _ in over_t.values()))
    return {"ok": bool(ok), "summary": summary, "costs": c, "tensors": t}


class _Ref:
    """Everything the CPU side contributes, computed ONCE for the module: models, batches, the oracle's matching, merge, eight
    updates (oneDNN on and off) and the fp64 anchor of the same sample."""


@pytest.fixture(scope="module")
def ref():
    from pleas_merging_amd import resnet as zoo
    from pleas_merging_amd.core.compiler import get_permutation_spec

    r = _Ref()
    threads = torch.get_num_threads()
    torch.set_num_threads(max(threads, min(16, _usable_cores())))
    try:
        # batch b = N(0, 1) seeded 1000 + b; matching takes batches 0..9, PLeaS batches 0..7
        r.data = [(torch.randn(BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(1000 + b)), None)
                  for b in range(N_MATCH)]
        models = []
        for seed in (0, 1):
            torch.manual_seed(seed)
            m = zoo.MODELS["resnet101"](num_classes=1000)
            zoo.calibrate_bn(m, [torch.randn(BATCH, 3, 224, 224, generator=torch.Generator().manual_seed(900 + i))
                                 for i in range(2)])
            models.append(m.eval())
        r.m1, r.m2 = models
        r.spec = get_permutation_spec(r.m1, ((1, 3, 224, 224),))
        # ---- oracle: matching on the same ten batches, merge + 8 updates from its assignment
        r.w = _Want()
        r.w.spec = r.spec
        r.w.want_perm, r.w.want_costs = orc.activation_matching(r.spec, r.m1, r.m2, r.data, N_MATCH, accumulate=True)
        o3 = orc.partial_merge(r.spec, r.m1, r.m2, r.w.want_perm, r.w.want_costs, 0.0)
        r.merged = {k: v.clone() for k, v in o3.state_dict().items()}
        o3, losses = orc.train(r.data[:N_UPDATES], r.m1, r.m2, o3, r.spec, r.w.want_perm, r.w.want_costs, 0.0, T_MAX)
        assert len(losses) == N_UPDATES
        r.want = {k: v.clone() for k, v in o3.state_dict().items()}
        with torch.backends.mkldnn.flags(enabled=False):      # the oracle against itself: the yardstick at this depth
            v3 = orc.partial_merge(r.spec, r.m1, r.m2, r.w.want_perm, r.w.want_costs, 0.0)
            v3, _ = orc.train(r.data[:N_UPDATES], r.m1, r.m2, v3, r.spec, r.w.want_perm, r.w.want_costs, 0.0, T_MAX)
        r.variant = v3.state_dict()
        # the fp64 ANCHOR: the same ten batches / eight updates in fp64 (oracle/pleas_oracle.fp64_anchor).  Distances to it are
        # statements about accuracy -- the oneDNN on / off yardstick above is ONE draw of the spread between two fp32 runs.
        r.costs64, r.want64 = orc.fp64_anchor(r.spec, r.m1, r.m2, r.data, N_MATCH, N_UPDATES, r.w.want_perm, r.w.want_costs, 0.0,
                                              T_MAX)
    finally:
        torch.set_num_threads(threads)
    return r


def test_rn101_timed_configuration_vs_oracle(ref):
    _hip_job_vs_reference(ref, "r05_timed_config")


def test_rn101_timed_configuration_split_bf16_vs_oracle(ref):
    """The same job under ``pleas_arith(PLEAS_ARITH_SPLIT_BF16)`` (matching contraction, k x k source convolutions, forward and
    weight gradient of the updates on bf16 MFMA with exactly split operands) against the SAME oracle results and the SAME gates,
    fp64 anchor included."""
    from pleas_merging_amd import _lib

    lib = _lib.lib()
    assert lib.pleas_arith_get() == 0
    lib.pleas_arith(1)
    try:
        _hip_job_vs_reference(ref, "r05_timed_config_split_bf16")
    finally:
        lib.pleas_arith(0)


def _hip_job_vs_reference(ref, tag):
    from pleas.methods.activation_matching import activation_matching
    from pleas.methods.partial_matching import partial_merge
    from pleas_merging_amd.methods.pleas_merging import FrozenSources, PleasFitter

    m1, m2, spec, data, w = ref.m1, ref.m2, ref.spec, ref.data, ref.w
    merged, want, variant, costs64, want64 = ref.merged, ref.want, ref.variant, ref.costs64, ref.want64
    g1, g2 = copy.deepcopy(m1).cuda(), copy.deepcopy(m2).cuda()
    gdata = [(x.cuda(), None) for x, _ in data]
    inputs = [x for x, _ in gdata[:N_UPDATES]]

    # ---- HIP path, as bench.run_job drives it
    early = {}

    def while_solving():
        early["sources"] = src = FrozenSources(g1, g2)
        early["taken"] = src.prefetch(inputs, group=8, max_groups=2, memory_fraction=0.7)

    perm, costs = activation_matching(spec, g1, g2, gdata, N_MATCH, output_costs=True, while_solving=while_solving,
                                      batches_per_forward=10)
    assert early["taken"] == 8, early["taken"]         # one 128-sample group was forwarded beside the LAP kernel
    flips = fs._check_matching(w, perm, costs)
    worst_cost = max(fs._rel(costs[k], w.want_costs[k]) for k in spec)
    print("timed configuration (%s), matching: worst group cost rel-fro %.2e, flipped groups %r" % (tag, worst_cost, flips))

    # ---- merge + 8 updates from the ORACLE's assignment (both sides merge the same blocks); the prefetched sources
    # do not depend on it
    gcosts = {k: v.cuda() for k, v in w.want_costs.items()}
    m3 = partial_merge(spec, g1, g2, w.want_perm, gcosts, 0.0, device=torch.device("cuda"))
    for k, v in m3.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.equal(v.cpu(), merged[k]), k
    fit = PleasFitter(g1, g2, m3, spec, w.want_perm, gcosts, 0.0, T_MAX, sources=early["sources"])
    n = 0
    for _ in fit.steps(inputs, sources_per_forward=8):
        n += 1
    assert n == N_UPDATES and fit.fast_updates == N_UPDATES - 1
    got = {k: v.cpu() for k, v in fit.finish().state_dict().items()}
    # ---- the gate.  A tensor's yardstick is ONE draw of "two correct implementations apart" (the oracle with oneDNN
    # convolutions on / off), and so is the HIP path's distance.  History of this comparison on the MI355X: with the vendor's
    # Winograd 3 x 3 kernels in the source forwards one tensor of 105 sat at 7 x its own yardstick (round 4); with the vendor's
    # direct kernels none above 3 x; since round 5 the k x k source convolutions are the library's own (repeatable bits): none above
    # 3 x here, one (layer3.11.conv2.weight) in the bench's model draw (profiles/r05_timed_config_parity.json, r05_a_bench.json).
    # The update's kernels themselves are held to fp64 on identical taps in test_hip_fullsize.py.  So, as for the long horizon
    # (tests/test_hip_long_horizon.py): at most two tensors of ~100 above 3 x their own yardstick, none above 3 x the model's
    # LARGEST yardstick; the share of coordinates with a visibly different Adam step within 3 x the oracle's worst share; all
    # other coordinates together within max(1e-4, 2 x the oracle's worst).  The fp64 anchor below is the statement about accuracy.
    rows = {}
    for k in want:
        if k == fs.DEGENERATE or not want[k].dtype.is_floating_point:
            continue
        if torch.equal(want[k], merged[k]):       # not trained (BatchNorm): must simply still be the merged tensor
            assert torch.equal(got[k], merged[k]), k
            continue
        r, yard = fs._rel(got[k], want[k]), fs._rel(variant[k], want[k])
        stats = []
        for other in (got[k], variant[k]):
            d = (other.double() - want[k].double()).abs()
            affected = d > fs.LR / 10
            stats.append((float(affected.double().mean()), float((d * ~affected).norm() / (want[k].double().norm() + 1e-30))))
        rows[k] = (r, yard) + stats[0] + stats[1]
    worst = max(rows, key=lambda k: rows[k][0])
    yard_max = max(v[1] for v in rows.values())
    over = {k: v[:2] for k, v in rows.items() if v[0] > max(fs.TOL, 3 * v[1])}
    above = {k: v[:2] for k, v in rows.items() if v[0] > fs.TOL}
    summary = {"batch": BATCH, "matching_batches_per_forward": N_MATCH, "sources_per_forward": N_UPDATES, "updates": N_UPDATES,
               "worst_group_cost_rel_fro": worst_cost, "flipped_groups": {k: list(v) for k, v in flips.items()},
               "tensors": len(rows), "worst_tensor": worst, "worst_rel_fro": rows[worst][0],
               "oracle_self_spread_of_that_tensor": rows[worst][1], "oracle_self_spread_worst": yard_max,
               "tensors_above_1e-4": len(above), "tensors_above_3x_own_yardstick": {k: list(v) for k, v in over.items()},
               "worst_share_of_flipped_adam_steps": max(v[2] for v in rows.values()),
               "oracle_worst_share_of_flipped_adam_steps": max(v[4] for v in rows.values()),
               "worst_rest_rel_fro": max(v[3] for v in rows.values()),
               "oracle_worst_rest_rel_fro": max(v[5] for v in rows.values()),
               "source_conv": __import__("pleas_merging_amd.methods.source_forward", fromlist=["x"]).SOURCE_CONV}
    print("timed configuration (%s), %d updates: %s" % (tag, N_UPDATES, json.dumps(summary, indent=1)))
    out_dir = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, tag + "_parity.json"), "w") as f:
            json.dump(summary, f, indent=1)
    assert len(over) <= max(2, len(rows) // 50), over      # 0 observed here, 0-2 in bench.py's lines; a wrong kernel moves dozens
    assert all(r <= max(fs.TOL, 3 * yard_max) for r, _ in over.values()), (over, yard_max)
    share_max, rest_max = max(v[4] for v in rows.values()), max(v[5] for v in rows.values())
    for k, (r, yard, frac, rest, yfrac, yrest) in rows.items():
        assert frac <= max(3 * share_max, fs.SHARE) and rest <= max(fs.TOL, 2 * rest_max), (k, rows[k], share_max, rest_max)
    gate_stem(got[fs.DEGENERATE], merged[fs.DEGENERATE], [want[fs.DEGENERATE], variant[fs.DEGENERATE]],
              lambda t: stem_objective(m1, m2, t, spec, w.want_perm, w.want_costs, 0.0, data[:N_UPDATES], 1000), what="stem")

    # ---- against the fp64 anchor: the HIP path must be as ACCURATE as the reference's fp32 arithmetic is
    anchor = fp64_gate({k: costs[k] for k in spec}, {k: w.want_costs[k] for k in spec}, costs64,
                       {k: got[k] for k in rows}, {k: want[k] for k in rows}, {k: variant[k] for k in rows}, want64)
    print("timed configuration (%s) vs the fp64 anchor: %s" % (tag, json.dumps(anchor["summary"], indent=1)))
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, tag + "_fp64_anchor.json"), "w") as f:
            json.dump(anchor, f, indent=1)
    assert anchor["ok"], anchor["summary"]
