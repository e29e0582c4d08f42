"""End-to-end parity of the four API functions on the MI355X against (a) the golden vectors
produced by the reference itself and (b) the CPU oracle.  Calls go through the drop-in
``pleas.*`` namespace, i.e. through libpleas_hip.so."""
import copy
import os

import pytest
import torch

from oracle import pleas_oracle as orc
from stem_gate import gate_stem, reference_stems, stem_objective

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-30))


def _cuda_pair(t):
    return copy.deepcopy(t.m1).cuda(), copy.deepcopy(t.m2).cuda()


@pytest.mark.parametrize("fx", ["tiny_basic", "tiny_bottleneck"])
def test_activation_matching_reference_mode_vs_golden(fx, request):
    from pleas.methods.activation_matching import activation_matching

    t = request.getfixturevalue(fx)
    m1, m2 = _cuda_pair(t)
    perm, costs = activation_matching(t.spec, m1, m2, t.batches(), 3, output_costs=True, accumulate="reference")
    want_p, want_c = t.per_key("am_perm"), t.per_key("am_cost")
    for k in t.spec:
        assert costs[k].is_cuda and perm[k].device.type == "cpu" and perm[k].dtype == torch.int64
        assert torch.allclose(costs[k].cpu(), want_c[k], rtol=1e-4, atol=1e-4), (k, _rel(costs[k], want_c[k]))
        assert (perm[k] == want_p[k]).all(), k


@pytest.mark.parametrize("fx", ["tiny_basic", "tiny_bottleneck"])
def test_activation_matching_accumulate_vs_oracle(fx, request):
    from pleas.methods.activation_matching import activation_matching, cross_features_inner_product

    t = request.getfixturevalue(fx)
    m1, m2 = _cuda_pair(t)
    for hip_fn, orc_fn in ((None, orc.cross_features_cdist), (cross_features_inner_product, orc.cross_features_inner_product)):
        kw = {} if hip_fn is None else {"cross_features": hip_fn}
        perm, costs = activation_matching(t.spec, m1, m2, t.batches(), 3, output_costs=True, **kw)
        want_p, want_c = orc.activation_matching(t.spec, t.m1, t.m2, t.batches(), 3, cross=orc_fn, accumulate=True)
        for k in t.spec:
            assert _rel(costs[k], want_c[k]) < 2e-5, k
            assert (perm[k] == want_p[k]).all(), k


def test_plug_points_generic_path(tiny_basic):
    """The HIP kernels used through the reference's own plug points (callables inside the fx
    graph, per-group solver), without the fused fast path."""
    from pleas.core.solvers import hip_solve_lsa
    from pleas.methods.activation_matching import build_cross_module, compute_matching_costs, cross_features_cdist

    t = tiny_basic
    m1, m2 = _cuda_pair(t)
    axes = [ax for g in t.spec.values() for ax in g.node]
    gm = build_cross_module(m1, m2, axes, lambda x, y, a: cross_features_cdist(x, y, a))
    costs = compute_matching_costs(t.spec, gm, t.batches(), 3, accumulate="reference")
    want_c = t.per_key("am_cost")
    for k in t.spec:
        assert torch.allclose(costs[k].cpu(), want_c[k], rtol=1e-4, atol=1e-4)
        assert (hip_solve_lsa(costs[k]) == t.per_key("am_perm")[k]).all()


def test_partial_merge_on_device_equals_reference_placement(tiny_basic):
    """``partial_merge(..., device=...)`` (addition) keeps the merged model on the GPU; the default is a CPU module as
    in the reference.  Same tensors either way, and the fitter accepts both."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    t = tiny_basic
    m1, m2 = _cuda_pair(t)
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    host = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
    dev = partial_merge(t.spec, m1, m2, perm, costs, 0.5, device="cuda")
    assert all(not p.is_cuda for p in host.parameters()) and all(p.is_cuda for p in dev.parameters())
    for (k, a), (_, b) in zip(host.state_dict().items(), dev.state_dict().items()):
        assert torch.equal(a, b.cpu()), k
    outs = []
    for m3 in (host, dev):
        fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, 3, num_classes=10)
        for x, _ in t.batches():
            fit.step(x)
        outs.append({k: v.cpu().clone() for k, v in fit.finish().state_dict().items()})
    for k in outs[0]:
        if k != DEGENERATE and outs[0][k].dtype.is_floating_point:
            assert _rel(outs[0][k], outs[1][k]) < 1e-6, k


def test_models_keep_mode_and_device(tiny_basic):
    from pleas.methods.activation_matching import activation_matching

    m1, m2 = _cuda_pair(tiny_basic)
    m1.train()
    activation_matching(tiny_basic.spec, m1, m2, tiny_basic.batches(), 1)
    assert m1.training and not m2.training and next(m1.parameters()).is_cuda


def test_cpu_models_fail_loudly(tiny_basic):
    from pleas.methods.activation_matching import activation_matching

    with pytest.raises(RuntimeError):
        activation_matching(tiny_basic.spec, tiny_basic.m1, tiny_basic.m2, tiny_basic.batches(), 1)


def test_weight_matching_vs_golden(tiny_basic):
    from pleas.methods.weight_matching import weight_matching

    t = tiny_basic
    sa = {k: v.cuda() for k, v in t.m1.state_dict().items()}
    sb = {k: v.cuda() for k, v in t.m2.state_dict().items()}
    perm, costs = weight_matching(t.spec, sa, sb, max_iter=100, seed=0, verbose=False, return_costs=True)
    for k in t.spec:
        assert (perm[k].cpu() == torch.from_numpy(t.z["wm_perm/%s" % k])).all(), k
        assert torch.allclose(costs[k].cpu(), torch.from_numpy(t.z["wm_cost/%s" % k]), rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("ratio", [0.0, 0.5, 1.0])
def test_blocks_and_partial_merge_vs_golden(tiny_basic, ratio):
    from pleas.methods.partial_matching import get_blocks, partial_merge

    t = tiny_basic
    tag = "r%03d" % int(ratio * 100)
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    blocks = get_blocks(t.spec, perm, costs, ratio, False)
    for k in t.spec:
        for j in range(4):
            assert (blocks[k][j].cpu() == torch.from_numpy(t.z["blocks_%s/%s/%d" % (tag, k, j)])).all(), (k, j)
    m1, m2 = _cuda_pair(t)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, ratio)
    want = t.state("merged_" + tag)
    got = m3.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert got[k].device.type == "cpu" and got[k].shape == want[k].shape, k
        assert torch.equal(got[k], want[k]), k  # gather / average / halve are exact in fp32
    assert not m3.training and not m3.conv1.weight.requires_grad


@pytest.mark.parametrize("ratio,steps", [(0.0, 5), (0.0, 20), (0.5, 5), (0.5, 20)])
def test_train_adam_vs_golden(tiny_basic, ratio, steps):
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    t = tiny_basic
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    m1, m2 = _cuda_pair(t)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, ratio)
    m3 = train(t.batches("xt"), m1, m2, m3, t.spec, perm, costs, ratio, False, steps, None, num_classes=10)
    want = t.state("trained_r%03d_s%d" % (int(ratio * 100), steps))
    got = m3.state_dict()
    # north-star tolerance: merged weights within 1e-4 rel-fro of the reference
    worst = max(_rel(got[k], want[k]) for k in want if want[k].dtype.is_floating_point and k != DEGENERATE)
    assert worst < 1e-4, worst
    # The stem's residual is rounding noise in the reference itself: gated on its layer objective and on the reference's
    # MEASURED self-disagreement (tests/stem_gate.py, tests/golden/stem_spread.npz), not on Adam's maximum travel.
    init, refs = reference_stems(ratio, steps)
    assert torch.equal(refs[0], want[DEGENERATE]) and torch.equal(init, t.state("merged_r%03d" % int(ratio * 100))[DEGENERATE])
    costs_c = t.per_key("am_cost")
    gate_stem(got[DEGENERATE], init, refs,
              lambda w: stem_objective(t.m1, t.m2, w, t.spec, perm, costs_c, ratio, t.batches("xt")[:steps + 1], 10),
              what="ratio %.1f, %d updates" % (ratio, steps + 1))


DEGENERATE = "conv1.weight"


def test_planted_permutation_is_recovered(tiny_bottleneck):
    """Known answer: model2 = model1 with every group permuted => matching returns the inverse
    permutation and a full merge (ratio 0) gives model1 back."""
    from pleas.core.utils import apply_perm, invert_perm, make_random_perm
    from pleas.methods.activation_matching import activation_matching
    from pleas.methods.partial_matching import partial_merge

    t = tiny_bottleneck
    m1 = copy.deepcopy(t.m1)
    planted = make_random_perm(t.spec, torch.Generator().manual_seed(2))
    m2 = copy.deepcopy(m1)
    apply_perm(planted, t.spec, m2, inplace=True)
    m1, m2 = m1.cuda(), m2.cuda()
    perm, costs = activation_matching(t.spec, m1, m2, t.batches(), 2, output_costs=True)
    inv = invert_perm(planted)
    for k in t.spec:
        assert (perm[k] == inv[k]).all(), k
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.0)
    for k, v in m1.state_dict().items():
        assert torch.allclose(m3.state_dict()[k], v.cpu(), rtol=1e-6, atol=1e-7), k


def test_spec_function_invariance_on_gpu(tiny_basic):
    from pleas.core.compiler import check_permutation_spec, get_permutation_spec

    m = copy.deepcopy(tiny_basic.m1).cuda()
    spec = get_permutation_spec(m, ((2, 3, 32, 32),))
    assert check_permutation_spec(m, spec, torch.randn(2, 3, 32, 32).cuda())


def test_grouped_equals_per_node_launches(tiny_bottleneck):
    from pleas.methods.activation_matching import activation_matching

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    pa, ca = activation_matching(t.spec, m1, m2, t.batches(), 3, output_costs=True, grouped=True)
    pb, cb = activation_matching(t.spec, m1, m2, t.batches(), 3, output_costs=True, grouped=False)
    for k in t.spec:
        assert _rel(ca[k], cb[k]) < 1e-6 and (pa[k] == pb[k]).all()


def test_two_stream_twin_forward_equals_interleaved(tiny_bottleneck):
    """Matching with model2's chain on a second HIP stream (default) vs the node-by-node interleaved twin graph."""
    from pleas_merging_amd import hip_ops
    from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    data = t.batches() + t.batches()
    main = torch.cuda.current_stream()
    a = accumulate_costs_fused(t.spec, m1, m2, data, 8, hip_ops.EPI_NEG_CDIST, overlap=False)
    b = accumulate_costs_fused(t.spec, m1, m2, data, 8, hip_ops.EPI_NEG_CDIST, overlap=True, fuse_bn=False)
    assert torch.cuda.current_stream() == main
    for k in t.spec:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("fx", ["tiny_basic", "tiny_bottleneck"])
def test_matching_with_fused_bn_chains_equals_module_chains(fx, request):
    """Twin forward with every eval-mode BatchNorm -> [+identity] -> [ReLU] chain as ONE launch that keeps all nodes
    (default) vs the vendor BN / add / ReLU modules: same costs up to the fp32 rounding of the fold, same matching."""
    from pleas_merging_amd import hip_ops
    from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused, solve_all
    from pleas_merging_amd.core.solvers import hip_solve_lsa

    t = request.getfixturevalue(fx)
    m1, m2 = _cuda_pair(t)
    data = t.batches()
    a = accumulate_costs_fused(t.spec, m1, m2, data, 4, hip_ops.EPI_NEG_CDIST, fuse_bn=False)
    b = accumulate_costs_fused(t.spec, m1, m2, data, 4, hip_ops.EPI_NEG_CDIST, fuse_bn=True, derive_bn=False)
    d = accumulate_costs_fused(t.spec, m1, m2, data, 4, hip_ops.EPI_NEG_CDIST, fuse_bn=True, derive_bn=True)
    for k in t.spec:
        assert _rel(a[k], b[k]) < 1e-5, k
        # BatchNorm nodes derived from the convolution node's products, norms and row sums instead of contracted
        assert _rel(d[k], b[k]) < 2e-5, k
    pa, pb, pd = solve_all(a, hip_solve_lsa), solve_all(b, hip_solve_lsa), solve_all(d, hip_solve_lsa)
    for k in t.spec:
        assert (pa[k] == pb[k]).all() and (pa[k] == pd[k]).all(), k
    for epi in (hip_ops.EPI_INNER,):
        u = accumulate_costs_fused(t.spec, m1, m2, data, 4, epi, fuse_bn=True, derive_bn=False)
        v = accumulate_costs_fused(t.spec, m1, m2, data, 4, epi, fuse_bn=True, derive_bn=True)
        for k in t.spec:
            assert _rel(v[k], u[k]) < 2e-5, k
    m1.train()   # training-mode BatchNorm is never folded: the module path must still work
    c = accumulate_costs_fused(t.spec, m1.eval(), m2, data, 4, hip_ops.EPI_NEG_CDIST, fuse_bn=True)
    for k in t.spec:
        assert torch.equal(d[k], c[k]), k


def _layer_objective(t, m3, ratio, perm, costs, batches, merging="perm_gradmask"):
    """Reference objective sum_layers mean((L(ip) - op)^2) summed over batches, on the CPU oracle."""
    blocks = orc.spread_blocks(t.spec, orc.get_blocks(t.spec, perm, costs, ratio))
    a1, a2 = {}, {}
    h = orc._hook_inputs(t.m1, a1) + orc._hook_inputs(t.m2, a2)
    total = 0.0
    with torch.no_grad():
        for x, _ in batches:
            t.m1(x)
            t.m2(x)
            for name, layer in m3.named_modules():
                if isinstance(layer, (torch.nn.Conv2d, torch.nn.Linear)):
                    ip, op = orc.layer_targets(orc.get_attr(t.m1, name.split(".")), orc.get_attr(t.m2, name.split(".")),
                                               blocks, name, a1[name], a2[name], num_classes=10, merging=merging)
                    total += float(((layer(ip) - op) ** 2).mean())
    for hh in h:
        hh.remove()
    return total


@pytest.mark.parametrize("ratio", [0.0, 0.5])
def test_train_normal_eq_minimises_the_reference_objective(tiny_basic, ratio):
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    t = tiny_basic
    perm, costs_c = t.per_key("am_perm"), t.per_key("am_cost")
    costs = {k: v.cuda() for k, v in costs_c.items()}
    data = t.batches("xt")[:21]
    m1, m2 = _cuda_pair(t)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, ratio)
    init = {k: v.clone() for k, v in m3.state_dict().items()}
    m_adam = train(data, m1, m2, copy.deepcopy(m3), t.spec, perm, costs, ratio, False, 20, None, num_classes=10)
    m_neq = train(data, m1, m2, m3, t.spec, perm, costs, ratio, False, 20, None, num_classes=10, solver="normal_eq")
    f_init = _layer_objective(t, orc.partial_merge(t.spec, t.m1, t.m2, perm, costs_c, ratio), ratio, perm, costs_c, data)
    f_adam = _layer_objective(t, m_adam.cpu(), ratio, perm, costs_c, data)
    f_neq = _layer_objective(t, m_neq.cpu(), ratio, perm, costs_c, data)
    assert f_neq <= f_adam * (1 + 1e-4) and f_neq < f_init, (f_init, f_adam, f_neq)
    if ratio == 0.0:
        # closed form of one layer against the fp64 oracle (normal equations + lstsq) on the same batches
        name = "layer2.0.conv2"
        layer = orc.get_attr(m_neq, name.split("."))
        blocks = orc.spread_blocks(t.spec, orc.get_blocks(t.spec, perm, costs_c, ratio))
        a1, a2 = {}, {}
        h = orc._hook_inputs(t.m1, a1) + orc._hook_inputs(t.m2, a2)
        ips, ops_ = [], []
        with torch.no_grad():
            for x, _ in data:
                t.m1(x)
                t.m2(x)
                ip, op = orc.layer_targets(orc.get_attr(t.m1, name.split(".")), orc.get_attr(t.m2, name.split(".")), blocks,
                                           name, a1[name], a2[name], num_classes=10)
                ips.append(ip)
                ops_.append(op)
        for hh in h:
            hh.remove()
        A, Bm = orc.normal_equations(ips, ops_, layer)
        want = orc.solve_normal_equations(A, Bm, ridge=1e-6).t().reshape(layer.weight.shape)
        assert _rel(layer.weight, want.float()) < 1e-3
    else:
        # frozen entries (reference gradient mask) keep their initial value
        from pleas.methods.partial_matching import get_blocks, spread_blocks
        from pleas.methods.pleas_merging import get_gradient_mask

        layers = {n: m for n, m in m_neq.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))}
        masks = get_gradient_mask(spread_blocks(t.spec, get_blocks(t.spec, perm, costs_c, ratio, False)), layers)
        k = 0
        for n, m in layers.items():
            for pn, p in m.named_parameters():
                frozen = masks[k] == 0
                assert torch.equal(p.detach().cpu()[frozen], init["%s.%s" % (n, pn)][frozen])
                k += 1


# ------------------------------------------------------------------------------------------ BASELINE.json configs at full size
def _rn(arch, seed, classes=1000):
    from pleas_merging_amd import resnet as zoo

    torch.manual_seed(seed)
    return zoo.MODELS[arch](num_classes=classes)


def test_config0_resnet18_weight_matching_vs_oracle():
    """configs[0]: ResNet-18 pair weight_matching, random-init weights, seed 0, max_iter 100 (reference driver
    run_domainnet.py:247-255): the HIP loop must take the same path (same LAP count, perms, costs) as the CPU oracle."""
    from pleas.core.compiler import get_permutation_spec
    from pleas.methods.weight_matching import weight_matching

    m1, m2 = _rn("resnet18", 0), _rn("resnet18", 1)
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    want_p, want_c, laps = orc.weight_matching(spec, m1.state_dict(), m2.state_dict(), 100, 0)
    sa = {k: v.cuda() for k, v in m1.state_dict().items()}
    sb = {k: v.cuda() for k, v in m2.state_dict().items()}
    perm, costs = weight_matching(spec, sa, sb, max_iter=100, seed=0, verbose=False, return_costs=True)
    assert laps >= len(spec)
    for k in spec:
        assert (perm[k].cpu() == want_p[k]).all(), k
        assert _rel(costs[k], want_c[k]) < 1e-5, k


def test_full_size_planted_permutation_resnet50():
    """Size-independent property at full scale (ResNet-50, 224x224, 37 groups up to 2048 wide): a planted
    permutation + small noise is recovered exactly and a full merge returns model 1 up to that noise."""
    from pleas.core.compiler import get_permutation_spec
    from pleas.core.utils import apply_perm, invert_perm, make_random_perm
    from pleas.methods.activation_matching import activation_matching
    from pleas_merging_amd import resnet as zoo

    m1 = _rn("resnet50", 0).cuda()
    gen = torch.Generator(device="cuda").manual_seed(5)
    data = [(torch.randn(8, 3, 224, 224, device="cuda", generator=gen), None) for _ in range(3)]
    zoo.calibrate_bn(m1, [d[0] for d in data])
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    planted = make_random_perm(spec, torch.Generator().manual_seed(2))
    m2 = copy.deepcopy(m1)
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1e-3 * p.abs().mean() * torch.randn(p.shape, device="cuda", generator=gen))
    apply_perm(planted, spec, m2, inplace=True)
    perm = activation_matching(spec, m1, m2, data, 2)
    inv = invert_perm(planted)
    for k in spec:
        assert (perm[k] == inv[k]).all(), k


def test_matching_several_batches_per_forward_equals_one_by_one(tiny_bottleneck):
    """``batches_per_forward``: k batches go through the twin forward as ONE forward of the concatenated batch, every
    tracked node (and every derived BatchNorm node) is still contracted per batch -- the distance epilogue is per batch
    (reference activation_matching.py:31-46, :123-134).  Same costs as one forward per batch up to the vendor kernels'
    rounding at another batch size, same assignments; 7 batches, so every group size leaves a smaller last forward.  Train
    mode (the drivers' mode): the fused chains fold BatchNorm on every batch's own samples, so concatenation is exact there
    too -- same costs, assignments, running statistics and counters as one forward per batch; with the vendor modules
    (``fuse_bn=False``) train-mode batches are never concatenated: bit-identical."""
    from pleas_merging_amd import hip_ops
    from pleas_merging_amd.core.solvers import hip_solve_lsa
    from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused, solve_all

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    data = (t.batches() + t.batches())[:7]
    data = [(x + 0.01 * i, y) for i, (x, y) in enumerate(data)]
    want = accumulate_costs_fused(t.spec, m1, m2, data, 7, hip_ops.EPI_NEG_CDIST, batches_per_forward=1)
    want_perm = solve_all(want, hip_solve_lsa)
    want = {k: v.clone() for k, v in want.items()}
    for per in (2, 3, 8, None):
        got = accumulate_costs_fused(t.spec, m1, m2, data, 7, hip_ops.EPI_NEG_CDIST, batches_per_forward=per)
        perm = solve_all(got, hip_solve_lsa)
        for k in t.spec:
            assert _rel(got[k], want[k]) < 1e-5, (per, k, _rel(got[k], want[k]))
            assert torch.equal(perm[k], want_perm[k]), (per, k)
    for m in (m1, m2):
        m.train()

    def restart():          # a pass moves the running statistics (not used in train mode): same start for the next one
        for m in (m1, m2):
            m.load_state_dict({k: v.cuda() for k, v in (t.m1 if m is m1 else t.m2).state_dict().items()})

    def stats():
        return {"%d.%s" % (i, k): v.clone() for i, m in enumerate((m1, m2)) for k, v in m.state_dict().items()
                if "running" in k or "num_batches" in k}

    a = accumulate_costs_fused(t.spec, m1, m2, data, 7, hip_ops.EPI_NEG_CDIST, batches_per_forward=1)
    a, a_perm, a_stats = {k: v.clone() for k, v in a.items()}, solve_all(a, hip_solve_lsa), stats()
    for per in (2, 4, None):
        restart()
        b = accumulate_costs_fused(t.spec, m1, m2, data, 7, hip_ops.EPI_NEG_CDIST, batches_per_forward=per)
        b_perm = solve_all(b, hip_solve_lsa)
        for k in t.spec:
            assert _rel(b[k], a[k]) < 1e-5, (per, k, _rel(b[k], a[k]))
            assert torch.equal(b_perm[k], a_perm[k]), (per, k)
        for k, v in stats().items():
            if "num_batches" in k:
                assert int(v) == int(a_stats[k]) == 7, k
            else:
                assert torch.allclose(v, a_stats[k], rtol=1e-5, atol=1e-6), (per, k)
    restart()
    c = accumulate_costs_fused(t.spec, m1, m2, data, 7, hip_ops.EPI_NEG_CDIST, batches_per_forward=1, fuse_bn=False)
    c = {k: v.clone() for k, v in c.items()}
    restart()
    d = accumulate_costs_fused(t.spec, m1, m2, data, 7, hip_ops.EPI_NEG_CDIST, batches_per_forward=4, fuse_bn=False)
    for k in t.spec:        # vendor BatchNorm modules in train mode: one batch per forward whatever is asked
        assert torch.equal(c[k], d[k]), k


def test_fused_source_forwards_match_module_forwards(tiny_bottleneck):
    """BN + add + ReLU folded into one HIP pass (default) vs the sources run module by module: same taps up to the
    fp32 rounding of the fold, same fitted weights within the training tolerance."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    data = t.batches()
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    outs, taps = [], []
    for fuse in (False, True):
        m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
        fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, 3, num_classes=10, fuse_sources=fuse)
        assert (fit.src1 is not m1) == fuse and (fit.src2 is not m2) == fuse
        fit._run_sources(data[0][0].cuda())
        torch.cuda.synchronize()   # the sources run on side streams
        taps.append(({k: v.clone() for k, v in fit.tap1.inputs.items()}, {k: v.clone() for k, v in fit.tap2.outputs.items()}))
        fit.tap1.clear()
        fit.tap2.clear()
        for x, _ in data:
            fit.step(x)
        outs.append({k: v.clone() for k, v in fit.finish().state_dict().items()})
    assert set(taps[0][0]) == set(taps[1][0]) and len(taps[0][0]) > 5
    for k in taps[0][0]:
        assert _rel(taps[0][0][k].cpu(), taps[1][0][k].cpu()) < 1e-5, k
        assert _rel(taps[0][1][k].cpu(), taps[1][1][k].cpu()) < 1e-5, k
    for k in outs[0]:
        if k == DEGENERATE:
            assert torch.allclose(outs[0][k], outs[1][k], atol=2 * 5e-4 * 4)
        elif outs[0][k].dtype.is_floating_point:
            assert _rel(outs[0][k], outs[1][k].cpu()) < 1e-4, k


def test_two_stream_source_forwards_equal_single_stream(tiny_bottleneck):
    """model2 on a second HIP stream (default) vs both sources on one stream: same kernels, same values."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    data = t.batches() + t.batches()
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    outs = []
    for overlap in (False, True):
        m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
        fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, 7, num_classes=10, overlap_sources=overlap)
        assert (fit._side_streams is not None) == overlap
        for x, _ in data:
            fit.step(x)
        outs.append({k: v.clone() for k, v in fit.finish().state_dict().items()})
    for k in outs[0]:
        if k == DEGENERATE:
            assert torch.allclose(outs[0][k], outs[1][k], atol=2 * 5e-4 * 8)
        elif outs[0][k].dtype.is_floating_point:
            # layers narrower than the HIP tile's 16 input channels (this 4-wide fixture) take the vendor's atomics
            # weight-gradient kernel, which is order-dependent run to run even on one stream: rounding only
            assert _rel(outs[0][k], outs[1][k]) < 1e-6, k


def test_lookahead_steps_equal_sequential_steps(tiny_bottleneck):
    """steps(): the next batch's source forwards are enqueued before the current update (two side streams, two tap
    generations in flight).  Same updates in the same order as step(x) one by one."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    xs = [x for x, _ in t.batches() + t.batches() + t.batches()]
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    outs = []
    for lookahead in (False, True):
        m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
        fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, len(xs) - 1, num_classes=10)
        if lookahead:
            assert list(fit.steps(xs, lookahead=True, sources_per_forward=1)) == list(range(len(xs)))
            assert not fit._queue
        else:
            for x in xs:
                fit.step(x)
        assert fit.step_count == len(xs)
        outs.append({k: v.clone() for k, v in fit.finish().state_dict().items()})
    for k in outs[0]:
        if k == DEGENERATE:
            assert torch.allclose(outs[0][k], outs[1][k], atol=2 * 5e-4 * len(xs))
        elif outs[0][k].dtype.is_floating_point:   # 4-wide layers use the vendor's atomics weight gradient: rounding only
            assert _rel(outs[0][k], outs[1][k]) < 1e-6, k
    # a prefetched batch must be the one the next call applies
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
    fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, 3, num_classes=10)
    fit.step(xs[0], next_x=xs[1])
    with pytest.raises(RuntimeError):
        fit.step(xs[2])
    fit.finish()


def test_paired_source_forwards_equal_one_forward_per_batch(tiny_bottleneck):
    """steps() runs the frozen sources once per GROUP of batches (one forward of the concatenated batch, the updates read
    their slices of its taps): same updates in the same order as one forward per batch, also with ragged tails."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import FrozenSources, PleasFitter

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    xs = [x for x, _ in t.batches() + t.batches() + t.batches()][:11]
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    outs = []
    for kw in ({"sources_per_forward": 1}, {}, {"sources_per_forward": 4}, {"sources_per_forward": 3},   # default: pairs
               {"lookahead": True}, {"sources_per_forward": 3, "lookahead": True},    # next group enqueued beforehand
               {"prefetch": 2}, {"prefetch": 1, "sources_per_forward": 3, "lookahead": True},
               {"prefetch": 3, "sources_per_forward": 1},
               # PINNED host batches: copied on the copy stream, the next group's copies started a group ahead (stage)
               {"pinned": True}, {"pinned": True, "sources_per_forward": 3, "prefetch": 1}, {"pinned": True, "sources_per_forward": 1}):
        m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
        kw = dict(kw)
        prefetch = kw.pop("prefetch", 0)
        batches = [x.pin_memory() for x in xs] if kw.pop("pinned", False) else xs
        sources = None
        if prefetch:      # the first groups' forwards are enqueued before the fitter exists (as during the LAP kernel)
            sources = FrozenSources(m1, m2)
            assert sources.prefetch(batches, group=kw.get("sources_per_forward"), max_groups=prefetch) == \
                prefetch * kw.get("sources_per_forward", 2)
        fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, len(xs) - 1, num_classes=10, sources=sources)
        assert list(fit.steps(batches, **kw)) == list(range(len(xs)))
        assert fit.step_count == len(xs) and not fit._queue and not fit.sources._staged
        outs.append({k: v.clone() for k, v in fit.finish().state_dict().items()})
    for other in outs[1:]:
        for k in outs[0]:
            if k == DEGENERATE:
                assert torch.allclose(outs[0][k], other[k], atol=2 * 5e-4 * len(xs))
            elif outs[0][k].dtype.is_floating_point:   # vendor kernels may round differently at another batch size
                assert _rel(outs[0][k], other[k]) < 1e-5, k


def test_replayed_updates_equal_layer_by_layer_updates(tiny_bottleneck):
    """From the second update of an input shape on, PleasFitter rewrites the tap addresses in the item tables of the
    grouped launches and launches again.  Same fitted weights as when every update goes layer by layer -- one by one,
    in groups, with a change of batch size in between (which falls back and records again)."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    xs = [x for x, _ in t.batches() + t.batches() + t.batches()][:10]
    xs = xs[:5] + [xs[5][:2]] + xs[6:]            # update 5 has 2 samples instead of 4
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    outs, fast = [], []
    for mode in ("layers", "singles", "groups"):
        m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
        fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, len(xs) - 1, num_classes=10)
        if mode == "layers":
            for x in xs:
                fit._replay = None                # never replay
                fit.step(x)
        elif mode == "singles":
            for x in xs:
                fit.step(x)
        else:
            assert list(fit.steps(xs, sources_per_forward=2)) == list(range(len(xs)))
        fast.append(fit.fast_updates)
        outs.append({k: v.clone() for k, v in fit.finish().state_dict().items()})
    assert fast[0] == 0 and fast[1] == len(xs) - 3 and fast[2] == len(xs) - 3   # three shape changes: 4 -> 2 -> 4
    for other in outs[1:]:
        for k in outs[0]:
            if k == DEGENERATE:
                assert torch.allclose(outs[0][k], other[k], atol=2 * 5e-4 * len(xs))
            elif outs[0][k].dtype.is_floating_point:
                assert _rel(outs[0][k], other[k]) < 1e-5, k


def test_config4_zip_budget_partial_merge_and_train_vs_oracle():
    """configs[4]-style partial merge at ResNet-18 scale: zip ratios (budget 1.55: stages 3-4 stay separate, which makes
    ODD merged widths 2n-1 = 511 / 1023), gradient masks with frozen blocks, two PLeaS updates.  HIP path vs CPU oracle."""
    from pleas.core.compiler import get_permutation_spec
    from pleas.methods.activation_matching import activation_matching
    from pleas.methods.extras import zip_ratios
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train
    from pleas_merging_amd import resnet as zoo

    m1, m2 = _rn("resnet18", 0, 50).eval(), _rn("resnet18", 1, 50).eval()
    g = torch.Generator().manual_seed(77)
    data = [(torch.randn(4, 3, 64, 64, generator=g), torch.zeros(4)) for _ in range(4)]
    zoo.calibrate_bn(m1, [d[0] for d in data])
    zoo.calibrate_bn(m2, [d[0] for d in data])
    spec = get_permutation_spec(m1, ((1, 3, 64, 64),))
    ratios = zip_ratios(spec, 1.55, (1.0, 1.24, 1.55, 1.71, 2.0))
    assert set(ratios.values()) == {0.0, 1.0}
    g1, g2 = copy.deepcopy(m1).cuda(), copy.deepcopy(m2).cuda()
    perm, costs = activation_matching(spec, g1, g2, data, 2, output_costs=True)
    want_p, want_c = orc.activation_matching(spec, m1, m2, data, 2, accumulate=True)
    for k in spec:
        assert (perm[k] == want_p[k]).all(), k
    m3 = partial_merge(spec, g1, g2, perm, costs, ratios)
    o3 = orc.partial_merge(spec, m1, m2, want_p, want_c, ratios)
    widths = {v.shape[0] for k, v in m3.state_dict().items() if k.endswith("conv1.weight")}
    assert 511 in widths or 1023 in widths          # the odd widths really occur
    for (k, a), (_, b) in zip(m3.state_dict().items(), o3.state_dict().items()):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-6, atol=1e-7), k
    m3 = train(data, g1, g2, m3, spec, perm, costs, ratios, False, 1, None, num_classes=50)
    o3, _ = orc.train(data, m1, m2, o3, spec, want_p, want_c, ratios, 1, num_classes=50)
    for (k, a), (_, b) in zip(m3.state_dict().items(), o3.state_dict().items()):
        if k != DEGENERATE and a.dtype.is_floating_point:
            assert _rel(a, b) < 1e-4, (k, _rel(a, b))


def test_steps_generator_leaves_the_callers_stream_current(tiny_bottleneck):
    """``PleasFitter.steps`` enters the fitter's stream per update: a consumer that breaks out of the loop early (or
    raises) is on its own stream again at once, ordered after the updates it saw -- not at some later garbage collection."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    t = tiny_bottleneck
    m1, m2 = _cuda_pair(t)
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    xs = [x for x, _ in t.batches() + t.batches()]
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
    fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, len(xs) - 1, num_classes=10)
    mine = torch.cuda.Stream()
    with torch.cuda.stream(mine):
        gen = fit.steps(xs)
        for i in gen:
            assert torch.cuda.current_stream() == mine          # also between two updates
            if i == 2:
                break
        assert torch.cuda.current_stream() == mine
        seen = fit.loss_sum.clone()                              # ordered after the three updates: no sync needed
        with pytest.raises(ZeroDivisionError):
            for i in fit.steps(xs[3:]):
                1 / 0
        assert torch.cuda.current_stream() == mine
    assert fit.step_count == 4 and float(seen.sum()) > 0
    assert torch.cuda.current_stream() == torch.cuda.default_stream()
    fit.finish()


def test_pinned_stream_handle_is_per_thread():
    """``hip_ops.pin_stream`` caches the raw stream handle for the calling thread only: another thread (a DataLoader /
    collate thread, a second fitter) keeps launching on ITS current stream, and scratch workspaces are per stream."""
    import threading

    from pleas_merging_amd import hip_ops

    side = torch.cuda.Stream()
    seen = {}

    def other():
        seen["handle"] = hip_ops._stream()
        seen["ws"] = hip_ops.Workspace.get(torch.device("cuda", 0))

    with torch.cuda.stream(side), hip_ops.pin_stream():
        assert hip_ops._stream() == side.cuda_stream
        with hip_ops.pin_stream():                                # nested pins restore the outer one
            pass
        assert hip_ops._stream() == side.cuda_stream
        th = threading.Thread(target=other)
        th.start()
        th.join()
        ws_side = hip_ops.Workspace.get(torch.device("cuda", 0))
    assert seen["handle"] == torch.cuda.default_stream().cuda_stream != side.cuda_stream
    assert seen["ws"] is not ws_side
    assert hip_ops._stream() == torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("fx", ["tiny_basic", "tiny_bottleneck"])
@pytest.mark.parametrize("modes", [(True, True), (True, False)])
def test_train_mode_matching_vs_oracle(fx, modes, request):
    """The reference drivers never call .eval() before activation_matching (run_domainnet.py:172-186, :257-264): BatchNorm
    then normalises with batch statistics and its running statistics move.  The fused / derived path folds those
    statistics per batch on the device (``pleas_bn_train_fold``): same costs and assignments as the oracle run in the same
    mode, same running statistics and batch counters afterwards, modes untouched.  Also one model in each mode."""
    from pleas.methods.activation_matching import activation_matching

    t = request.getfixturevalue(fx)
    cpu = [copy.deepcopy(t.m1).train(modes[0]), copy.deepcopy(t.m2).train(modes[1])]
    gpu = [copy.deepcopy(m).cuda() for m in cpu]
    data = t.batches()
    want_p, want_c = orc.activation_matching(t.spec, cpu[0], cpu[1], data, 3, accumulate=True)
    perm, costs = activation_matching(t.spec, gpu[0], gpu[1], data, 3, output_costs=True)
    for k in t.spec:
        assert _rel(costs[k], want_c[k]) < 2e-5, (k, _rel(costs[k], want_c[k]))
        assert (perm[k] == want_p[k]).all(), k
    for g, c, mode in zip(gpu, cpu, modes):
        assert g.training == mode
        for (k, a), (_, b) in zip(g.state_dict().items(), c.state_dict().items()):
            if "running_" in k:
                assert torch.allclose(a.cpu(), b, rtol=1e-5, atol=1e-6), k
                assert mode or torch.equal(a.cpu(), dict(t.m2.state_dict())[k])      # eval mode: untouched
            elif k.endswith("num_batches_tracked"):
                assert int(a) == int(b) == (3 if mode else 0), k
    # the module-by-module path (vendor BatchNorm kernels) gives the same costs
    from pleas_merging_amd import hip_ops
    from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused

    again = [copy.deepcopy(m).cuda() for m in (t.m1.train(modes[0]), t.m2.train(modes[1]))]
    t.m1.eval(), t.m2.eval()
    plain = accumulate_costs_fused(t.spec, again[0], again[1], data, 3, hip_ops.EPI_NEG_CDIST, fuse_bn=False)
    for k in t.spec:
        assert _rel(plain[k], costs[k]) < 2e-5, k


def test_normal_eq_conv_layers_with_bias(tiny_basic):
    """solver="normal_eq" on convolutions WITH bias: the bias is one more column of ones in U (column sums of im2col(ip),
    of the target, and the row count join A and B).  Weight and bias of a 3x3 and a strided 1x1 convolution against the
    fp64 oracle (normal equations + lstsq) on the same batches."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    t = tiny_basic
    g = torch.Generator().manual_seed(3)
    pair = [copy.deepcopy(t.m1), copy.deepcopy(t.m2)]
    names = ("layer1.0.conv2", "layer2.0.downsample.0")
    for m in pair:
        for n in names:
            conv = orc.get_attr(m, n.split("."))
            conv.bias = torch.nn.Parameter(0.3 * torch.randn(conv.out_channels, generator=g))
    perm, costs_c = t.per_key("am_perm"), t.per_key("am_cost")
    costs = {k: v.cuda() for k, v in costs_c.items()}
    data = t.batches("xt")[:21]
    m1, m2 = (copy.deepcopy(m).cuda() for m in pair)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.0)
    m3 = train(data, m1, m2, m3, t.spec, perm, costs, 0.0, False, 20, None, num_classes=10, solver="normal_eq")
    blocks = orc.spread_blocks(t.spec, orc.get_blocks(t.spec, perm, costs_c, 0.0))
    a1, a2 = {}, {}
    h = orc._hook_inputs(pair[0], a1) + orc._hook_inputs(pair[1], a2)
    stacks = {n: ([], []) for n in names}
    with torch.no_grad():
        for x, _ in data:
            pair[0](x)
            pair[1](x)
            for n in names:
                ip, op = orc.layer_targets(orc.get_attr(pair[0], n.split(".")), orc.get_attr(pair[1], n.split(".")), blocks, n,
                                           a1[n], a2[n], num_classes=10)
                stacks[n][0].append(ip)
                stacks[n][1].append(op)
    for hh in h:
        hh.remove()
    for n in names:
        layer = orc.get_attr(m3, n.split("."))
        assert layer.bias is not None
        A, Bm = orc.normal_equations(stacks[n][0], stacks[n][1], layer)
        sol = orc.solve_normal_equations(A, Bm, ridge=1e-6)              # (K + 1, Cout), last row = bias
        want_w = sol[:-1].t().reshape(layer.weight.shape)
        assert _rel(layer.weight, want_w.float()) < 2e-3, (n, _rel(layer.weight, want_w.float()))
        assert _rel(layer.bias, sol[-1].float()) < 2e-3, (n, _rel(layer.bias, sol[-1].float()))


def _separatels_gate(t, perm, costs_c, ratio, got, want, init, data, lr_travel=5e-4):
    """perm_separatels stacks [i11, i1c, 0] -> [o11, o1c, 0] and [i22, 0, i2c] -> [o22, 0, o2c] (pleas_merging.py:132-137).
    For the rows of model 1's separate units the first half reproduces its target EXACTLY (model 1's own rows on model 1's
    own inputs) and the second half feeds zeros into the [separate-1 x separate-1] columns: that block -- and
    [separate-2 x separate-2] likewise -- has a gradient of exactly zero in real arithmetic (fp64: 1e-18 against 1e-2 in
    the other blocks).  As for the stem (tests/stem_gate.py), Adam integrates the convolution kernels' rounding noise
    there; after the first +-lr steps those rows no longer reproduce their target, so the noise reaches every column of
    the SEPARATE rows through the residual.  Gate: the merged rows (unaffected) meet the north-star tolerance; the
    separate rows stay within the reference's own travel from the merged value; and every layer's objective -- the
    reference's loss on the same batches -- equals the reference-trained layer's to 1e-3."""
    from pleas_merging_amd.core.utils import Axis

    blocks = orc.spread_blocks(t.spec, orc.get_blocks(t.spec, perm, costs_c, ratio))
    checked = 0
    for k in want:
        if k == DEGENERATE or not want[k].dtype.is_floating_point:
            continue
        a, b = got[k].double().cpu(), want[k].double()
        bo = blocks.get(Axis(k, 0))
        if b.dim() >= 2 and k.endswith(".weight") and bo is not None and len(bo[2]) > 0:
            no = len(bo[0])
            assert _rel(a[:no], b[:no]) < 1e-4, (k, _rel(a[:no], b[:no]))
            travel_ref = float((b[no:] - init[k].double()[no:]).abs().max())
            travel_got = float((a[no:] - init[k].double()[no:]).abs().max())
            assert travel_got <= 1.5 * travel_ref + lr_travel, (k, travel_got, travel_ref)
            checked += 1
        elif b.dim() >= 2:
            assert _rel(a, b) < 1e-4, (k, _rel(a, b))
    assert checked >= 5
    # objective of every layer under either set of weights
    a1, a2 = {}, {}
    hooks = orc._hook_inputs(t.m1, a1) + orc._hook_inputs(t.m2, a2)
    names = sorted({k.rsplit(".", 1)[0] for k in want if k.endswith(".weight") and want[k].dim() >= 2} - {"conv1"})
    totals = {n: [0.0, 0.0] for n in names}
    with torch.no_grad():
        for x, _ in data:
            t.m1(x)
            t.m2(x)
            for n in names:
                l1, l2 = orc.get_attr(t.m1, n.split(".")), orc.get_attr(t.m2, n.split("."))
                ip, op = orc.layer_targets(l1, l2, blocks, n, a1[n], a2[n], num_classes=10, merging="perm_separatels")
                for j, sd in enumerate((got, want)):
                    w, bias = sd[n + ".weight"].float().cpu(), sd.get(n + ".bias")
                    bias = bias.float().cpu() if bias is not None else None
                    out = torch.nn.functional.conv2d(ip, w, bias, l1.stride, l1.padding) if w.dim() == 4 \
                        else torch.nn.functional.linear(ip, w, bias)
                    totals[n][j] += float(((out - op) ** 2).mean())
    for hh in hooks:
        hh.remove()
    for n, (f_got, f_ref) in totals.items():
        assert abs(f_got - f_ref) <= 1e-3 * f_ref, (n, f_got, f_ref)


@pytest.mark.parametrize("mode,ratio", [("reg_mean", 0.0), ("perm_separatels", 0.5), ("perm_mixedls", 0.5)])
@pytest.mark.parametrize("steps", [5, 20])
def test_train_other_merging_modes_vs_golden(tiny_basic, mode, ratio, steps):
    """The reference's other ``merging`` modes (pleas_merging.py:125-144: two half-batches stacked along the sample axis,
    zeros in the absent blocks) on the HIP path -- two entries per layer in the grouped merge / forward / weight-gradient
    launches, gradients summed -- against weights trained by the reference itself (tests/golden/tiny_modes.npz)."""
    import numpy as np
    from conftest import GOLDEN
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    t = tiny_basic
    z = np.load(os.path.join(GOLDEN, "tiny_modes.npz"))
    perm = t.per_key("am_perm")
    costs = {k: v.cuda() for k, v in t.per_key("am_cost").items()}
    m1, m2 = _cuda_pair(t)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, ratio)
    init = {k: v.clone() for k, v in m3.state_dict().items()}
    m3 = train(t.batches("xt"), m1, m2, m3, t.spec, perm, costs, ratio, False, steps, None, merging=mode, num_classes=10)
    tag = "trained_%s_r%03d_s%d/" % (mode, int(ratio * 100), steps)
    got = m3.state_dict()
    want_all = {k: torch.from_numpy(z[tag + k]) for k in got}
    if mode == "perm_separatels":
        _separatels_gate(t, perm, t.per_key("am_cost"), ratio, got, want_all, init, t.batches("xt")[:steps + 1])
    else:
        worst = 0.0
        for k in got:
            if want_all[k].dtype.is_floating_point and k != DEGENERATE:
                worst = max(worst, _rel(got[k], want_all[k]))
        assert worst < 1e-4, (mode, worst)
    # a substring selects the stacked modes in the reference (`'perm_mixedls' in merging`)
    if mode != "reg_mean" and steps == 5:
        again = partial_merge(t.spec, m1, m2, perm, costs, ratio)
        again = train(t.batches("xt"), m1, m2, again, t.spec, perm, costs, ratio, False, steps, None,
                      merging=mode + "_v2", num_classes=10)
        if mode == "perm_mixedls":      # (the noise blocks of perm_separatels differ run to run where vendor kernels do)
            for k in got:
                if got[k].dtype.is_floating_point and k != DEGENERATE:
                    assert _rel(again.state_dict()[k], got[k]) < 1e-5, k


def test_other_merging_modes_bottleneck_vs_oracle(tiny_bottleneck):
    """Same modes on the bottleneck fixture (1x1 and strided layers, residual stream) against the CPU oracle, through the
    replayed-table fast path (more updates than shapes)."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    t = tiny_bottleneck
    perm, costs_c = t.per_key("am_perm"), t.per_key("am_cost")
    costs = {k: v.cuda() for k, v in costs_c.items()}
    data = (t.batches() + t.batches())[:7]
    for mode, ratio in (("perm_separatels", 0.5), ("perm_mixedls", 0.5), ("reg_mean", 0.0)):
        m1, m2 = _cuda_pair(t)
        m3 = partial_merge(t.spec, m1, m2, perm, costs, ratio)
        init = {k: v.clone() for k, v in m3.state_dict().items()}
        m3 = train(data, m1, m2, m3, t.spec, perm, costs, ratio, False, 6, None, merging=mode, num_classes=10)
        o3 = orc.partial_merge(t.spec, t.m1, t.m2, perm, costs_c, ratio)
        o3, _ = orc.train(data, t.m1, t.m2, o3, t.spec, perm, costs_c, ratio, 6, num_classes=10, merging=mode)
        if mode == "perm_separatels":
            _separatels_gate(t, perm, costs_c, ratio, m3.state_dict(), o3.state_dict(), init, data)
            continue
        for (k, a), (_, b) in zip(m3.state_dict().items(), o3.state_dict().items()):
            if k != DEGENERATE and a.dtype.is_floating_point:
                assert _rel(a, b) < 1e-4, (mode, k, _rel(a, b))


@pytest.mark.parametrize("mode,ratio", [("perm_separatels", 0.0), ("perm_mixedls", 0.0), ("reg_mean", 0.0),
                                        ("perm_separatels", 0.5), ("perm_mixedls", 0.5)])
def test_normal_eq_stacked_modes_vs_oracle(tiny_basic, mode, ratio):
    """solver="normal_eq" on the stacked objectives (two half-batches under the same weights, reference :125-145): A and
    B are the sums over both halves.  Every fully free layer against the fp64 oracle (normal equations of the stacked
    ``layer_targets`` + lstsq) on the same batches, and the reference objective no higher than Adam's."""
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train

    t = tiny_basic
    perm, costs_c = t.per_key("am_perm"), t.per_key("am_cost")
    costs = {k: v.cuda() for k, v in costs_c.items()}
    data = t.batches("xt")[:21]
    m1, m2 = _cuda_pair(t)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, ratio)
    m_adam = train(data, m1, m2, copy.deepcopy(m3), t.spec, perm, costs, ratio, False, 20, None, num_classes=10, merging=mode)
    m_neq = train(data, m1, m2, m3, t.spec, perm, costs, ratio, False, 20, None, num_classes=10, merging=mode,
                  solver="normal_eq")
    f_adam = _layer_objective(t, m_adam.cpu(), ratio, perm, costs_c, data, merging=mode)
    f_neq = _layer_objective(t, m_neq.cpu(), ratio, perm, costs_c, data, merging=mode)
    assert f_neq <= f_adam * (1 + 1e-4), (mode, f_adam, f_neq)
    blocks = orc.spread_blocks(t.spec, orc.get_blocks(t.spec, perm, costs_c, ratio))
    from pleas.methods.partial_matching import get_blocks, spread_blocks
    from pleas.methods.pleas_merging import get_gradient_mask

    layers = {n: m for n, m in m_neq.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))}
    masks = get_gradient_mask(spread_blocks(t.spec, get_blocks(t.spec, perm, costs_c, ratio, False)), layers)
    free, k = {}, 0
    for n, m in layers.items():
        free[n] = bool((masks[k] != 0).all())
        k += len(list(m.named_parameters()))
    a1, a2 = {}, {}
    h = orc._hook_inputs(t.m1, a1) + orc._hook_inputs(t.m2, a2)
    ips, ops_ = {n: [] for n in layers}, {n: [] for n in layers}
    with torch.no_grad():
        for x, _ in data:
            t.m1(x)
            t.m2(x)
            for n in layers:
                ip, op = orc.layer_targets(orc.get_attr(t.m1, n.split(".")), orc.get_attr(t.m2, n.split(".")), blocks, n,
                                           a1[n], a2[n], num_classes=10, merging=mode)
                ips[n].append(ip)
                ops_[n].append(op)
    for hh in h:
        hh.remove()
    checked = 0
    for n, layer in layers.items():
        if n == "conv1" or not free[n] or getattr(layer, "bias", None) is not None:
            continue
        A, Bm = orc.normal_equations(ips[n], ops_[n], layer)
        want = orc.solve_normal_equations(A, Bm, ridge=1e-6).t().reshape(layer.weight.shape)
        assert _rel(layer.weight, want.float()) < 1e-3, (mode, n, _rel(layer.weight, want.float()))
        checked += 1
    assert checked >= (3 if ratio == 0.0 else 0), (mode, checked)      # partial merges freeze blocks of most layers
