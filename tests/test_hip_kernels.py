"""Kernel-level parity: every C-ABI entry point against the CPU oracle / scipy / torch fp32.

All tests here need the MI355X (`-m gpu`) and go through libpleas_hip.so.
"""
import copy

import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from oracle import pleas_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from pleas_merging_amd import hip_ops

    return hip_ops


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


GRAM_SHAPES = [
    # (shape, axis)                       what it exercises
    ((4, 8, 6, 6), 1),      # tiny C, HW % 4 == 0, masked rows
    ((3, 20, 7, 7), 1),     # HW = 49: scalar-load path, chunk straddles samples
    ((2, 64, 28, 28), 1),   # one 64-tile, vector path
    ((2, 96, 14, 14), 1),   # 128-tile with masked rows
    ((16, 256, 14, 14), 1), # the most common ResNet-101 node
    ((5, 130, 9, 9), 1),    # 2x2 tiles, ragged everything, HW = 81
    ((16, 512, 1, 1), 1),   # avgpool-like (B, C, 1, 1)
    ((16, 300), 1),         # flatten-like 2-D node
    ((64, 32, 3, 3), 0),    # conv weight, axis 0 (B = 1)
    ((64, 32, 3, 3), 1),    # conv weight, axis 1 (B = Cout, HW = 9)
    ((10, 48), 0),          # fc weight axis 0
    ((48,), 0),             # BN vector: K = 1
    ((2, 64, 112, 112), 1), # long K, split-K across many workgroups
]


@pytest.mark.parametrize("shape,axis", GRAM_SHAPES)
def test_gram_inner_and_cdist(ops, shape, axis):
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    x = torch.randn(shape, generator=g)
    y = 0.7 * x + 0.5 * torch.randn(shape, generator=g)
    xd, yd = x.cuda(), y.cuda()
    inner = ops.cross_features_inner_product(xd, yd, axis).cpu()
    want = orc.cross_features_inner_product(x.double(), y.double(), axis)
    assert _rel(inner, want) < 2e-6
    dist = ops.cross_features_cdist(xd, yd, axis).cpu()
    want64 = orc.cross_features_cdist_f64(x, y, axis)
    ref32 = orc.cross_features_cdist(x, y, axis)
    # tolerance = the fp32 reference formula's own distance from fp64 (x3) + a floor
    tol = max(3 * _rel(ref32, want64), 2e-6)
    assert _rel(dist, want64) < tol, (_rel(dist, want64), tol)


def test_gram_accumulates_and_overwrites(ops):
    x, y = torch.randn(2, 40, 5, 5), torch.randn(2, 40, 5, 5)
    acc = torch.full((40, 40), 3.0, device="cuda")
    ops.gram_accum(x.cuda(), y.cuda(), 1, acc, ops.EPI_INNER, accumulate=True)
    want = 3.0 + orc.cross_features_inner_product(x, y, 1)
    assert torch.allclose(acc.cpu(), want, rtol=1e-5, atol=1e-5)
    ops.gram_accum(x.cuda(), y.cuda(), 1, acc, ops.EPI_INNER, accumulate=False)
    assert torch.allclose(acc.cpu(), want - 3.0, rtol=1e-5, atol=1e-5)


def test_gram_deterministic(ops):
    x, y = torch.randn(8, 256, 14, 14).cuda(), torch.randn(8, 256, 14, 14).cuda()
    a = ops.cross_features_cdist(x, y, 1)
    b = ops.cross_features_cdist(x, y, 1)
    assert torch.equal(a, b)


def test_gram_identical_inputs_give_zero_diagonal_floor(ops):
    # x == y: |x|^2 + |y|^2 - 2 x.y cancels to rounding noise; clamp keeps sqrt real
    x = torch.randn(4, 64, 8, 8).cuda()
    d = ops.cross_features_cdist(x, x, 1)
    assert torch.isfinite(d).all() and (d <= 0).all()
    assert d.diagonal().abs().max() < 0.05 * d.abs().mean()


def test_gram_rejects_cpu_tensors(ops):
    with pytest.raises(ops.PleasHipError):
        ops.cross_features_cdist(torch.randn(2, 4, 3, 3), torch.randn(2, 4, 3, 3), 1)


# ------------------------------------------------------------------------------------------ LAP
def _lap_cases(n, rng):
    x = rng.standard_normal((n, 32)).astype(np.float32)
    y = (x[rng.permutation(n)] + 0.1 * rng.standard_normal((n, 32))).astype(np.float32)
    return {
        "normal": rng.standard_normal((n, n)).astype(np.float32),
        "ties": rng.integers(0, 3, (n, n)).astype(np.float32),
        "binary": rng.integers(0, 2, (n, n)).astype(np.float32),
        "equal": np.zeros((n, n), np.float32),
        "cdist": -np.sqrt(np.maximum(((x[:, None] - y[None]) ** 2).sum(-1), 0)).astype(np.float32),
    }


@pytest.mark.parametrize("maximize", [True, False])
def test_lsap_bit_exact_vs_scipy_batched(ops, maximize):
    rng = np.random.default_rng(7)
    mats = []
    for n in (1, 2, 3, 5, 17, 64, 100, 255, 256, 257, 300, 512):
        mats += list(_lap_cases(n, rng).values())
    outs = ops.solve_lsa_batched([torch.from_numpy(m).cuda() for m in mats], maximize=maximize)
    for m, o in zip(mats, outs):
        _, want = linear_sum_assignment(m, maximize=maximize)
        assert (o.cpu().numpy() == want).all(), (m.shape, maximize)


@pytest.mark.parametrize("n", [1024, 2048, 2049, 3000, 4096])
def test_lsap_large(ops, n):
    rng = np.random.default_rng(n)
    cases = _lap_cases(n, rng)
    mats = [cases["normal"], cases["cdist"], cases["ties"]]
    outs = ops.solve_lsa_batched([torch.from_numpy(m).cuda() for m in mats], maximize=True)
    for m, o in zip(mats, outs):
        _, want = linear_sum_assignment(m, maximize=True)
        assert (o.cpu().numpy() == want).all()


def test_lsap_golden_fixture(ops):
    import os
    from conftest import GOLDEN

    z = np.load(os.path.join(GOLDEN, "lap_small.npz"))
    idx = list(range(int(z["n_cases"])))
    for mx, tag in ((True, "max"), (False, "min")):
        outs = ops.solve_lsa_batched([torch.from_numpy(z["cost_%d" % i]).cuda() for i in idx], maximize=mx)
        for i, o in zip(idx, outs):
            assert (o.cpu().numpy() == z["col_%s_%d" % (tag, i)]).all(), (i, tag)


def test_hip_solve_lsa_signature(ops):
    a = torch.randn(33, 33)
    got = ops.hip_solve_lsa(a.cuda())
    assert got.device.type == "cpu" and got.dtype == torch.int64
    assert (got == orc.solve_lsa(a)).all()


def test_lsap_rejects_bad_sizes(ops):
    with pytest.raises(ops.PleasHipError):
        ops.solve_lsa_batched([torch.zeros(3, 4).cuda()])
    with pytest.raises(ops.PleasHipError):
        ops.solve_lsa_batched([torch.zeros(4097, 4097).cuda()])


# ------------------------------------------------------------------------------------------ merge / adam / sqerr
def test_merge_blocks_two_axis(ops):
    g = torch.Generator().manual_seed(3)
    W1, W2 = torch.randn(12, 10, 3, 3, generator=g), torch.randn(12, 10, 3, 3, generator=g)
    bo = (torch.tensor([0, 3, 5, 7, 9, 11, 1]), torch.tensor([2, 0, 4, 6, 8, 10, 3]), torch.tensor([2, 4, 6, 8, 10]),
          torch.tensor([1, 5, 7, 9, 11]))
    bi = (torch.tensor([1, 2, 3, 4]), torch.tensor([4, 3, 2, 1]), torch.tensor([0, 5, 6, 7, 8, 9]),
          torch.tensor([0, 5, 6, 7, 8, 9]))
    from pleas_merging_amd.methods.partial_matching import block_maps

    r1, r2, nm = block_maps(bo, "cuda")
    c1, c2, _ = block_maps(bi, "cuda")
    got = ops.merge_blocks(W1.cuda(), W2.cuda(), 0, r1, r2, nm, c1, c2).cpu()
    from pleas_merging_amd.core.utils import Axis, PermutationGroup

    spec = {Axis("w", 0): PermutationGroup(12, {Axis("w", 0), Axis("d", 0)}, set()),
            Axis("w", 1): PermutationGroup(10, {Axis("w", 1), Axis("e", 0)}, set())}
    want = orc.merged_state(spec, {"w": W1}, {"w": W2}, {Axis("w", 0): bo, Axis("w", 1): bi})["w"]
    assert torch.equal(got, want)


def test_merge_blocks_one_axis_activation(ops):
    g = torch.Generator().manual_seed(4)
    x1, x2 = torch.randn(3, 9, 5, 5, generator=g), torch.randn(3, 9, 5, 5, generator=g)
    b = (torch.tensor([0, 2, 4, 6]), torch.tensor([1, 3, 5, 7]), torch.tensor([1, 3, 5, 7, 8]),
         torch.tensor([0, 2, 4, 6, 8]))
    from pleas_merging_amd.methods.partial_matching import block_maps

    r1, r2, nm = block_maps(b, "cuda")
    got = ops.merge_blocks(x1.cuda(), x2.cuda(), 1, r1, r2, nm).cpu()
    want = torch.cat([(x1[:, b[0]] + x2[:, b[1]]) / 2, x1[:, b[2]], x2[:, b[3]]], 1)
    assert torch.equal(got, want)


@pytest.mark.parametrize("shape,axis", [((16, 256, 14, 14), 1), ((2, 64, 28, 28), 1), ((5, 132, 6, 6), 1), ((2, 64, 112, 112), 1)])
def test_gram_split_bf16_study_switch(ops, shape, axis):
    """The STUDY arithmetic of the contraction (three-way bf16 split, six bf16 MFMAs per fp32 product; off by default,
    DESIGN.md): within the exact path's own distance from fp64 (x2 + a floor) on 16-byte-loadable shapes, and the switch
    really goes back to the exact path (bit-identical results before and after)."""
    from pleas_merging_amd import _lib

    g = torch.Generator().manual_seed(77)
    x = torch.randn(shape, generator=g)
    y = 0.7 * x + 0.5 * torch.randn(shape, generator=g)
    xd, yd = x.cuda(), y.cuda()
    want = orc.cross_features_inner_product(x.double(), y.double(), axis)
    exact = ops.cross_features_inner_product(xd, yd, axis).cpu()
    try:
        _lib.lib().pleas_gram_split_bf16(1)
        study = ops.cross_features_inner_product(xd, yd, axis).cpu()
        study_d = ops.cross_features_cdist(xd, yd, axis).cpu()
    finally:
        _lib.lib().pleas_gram_split_bf16(0)
    assert not torch.equal(study, exact)          # another arithmetic did run
    assert _rel(study, want) < max(2 * _rel(exact, want), 5e-7), (_rel(study, want), _rel(exact, want))
    assert _rel(study_d, orc.cross_features_cdist_f64(x, y, axis)) < 2e-6
    assert torch.equal(ops.cross_features_inner_product(xd, yd, axis).cpu(), exact)


@pytest.mark.parametrize("shape", [(3, 16, 8, 8), (2, 7, 7, 7), (4, 64, 14, 14), (1, 5, 3, 1)])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (False, False)])
def test_bn_act_matches_eval_batchnorm(ops, shape, with_res, relu):
    """Folded inference BN (+ residual) (+ ReLU) vs the torch fp32 module chain (tolerance: fp32 rounding of the fold)."""
    g = torch.Generator().manual_seed(31)
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C).eval()
    bn.running_mean.copy_(torch.randn(C, generator=g))
    bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn.weight.data.copy_(torch.rand(C, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(C, generator=g))
    x, res = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    with torch.no_grad():
        want = bn(x) + (res if with_res else 0)
        want = torch.relu(want) if relu else want
    scale = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
    shift = (bn.bias.double() - bn.running_mean.double() * scale).float().cuda()
    got = ops.bn_act(x.cuda(), scale.float().cuda(), shift, res.cuda() if with_res else None, relu).cpu()
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    if relu:
        assert (got >= 0).all()


@pytest.mark.parametrize("shape", [(3, 16, 8, 8), (2, 7, 7, 7)])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (True, False), (False, False)])
def test_bn_act_tracked_keeps_every_node_of_the_chain(ops, shape, with_res, relu):
    g = torch.Generator().manual_seed(32)
    C = shape[1]
    scale, shift = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    x, res = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
    bn, sm, act = ops.bn_act_tracked(x, scale, shift, res if with_res else None, relu)
    want_bn = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    assert torch.allclose(bn, want_bn, rtol=1e-6, atol=1e-6)
    assert (sm is None) == (not with_res) and (act is None) == (not relu)
    last = bn
    if with_res:
        assert torch.equal(sm, bn + res)
        last = sm
    if relu:
        assert torch.equal(act, torch.relu(last))
    # the single-output form is the same pass without the extra stores
    assert torch.equal(ops.bn_act(x, scale, shift, res if with_res else None, relu), act if relu else last)


@pytest.mark.parametrize("with_res,relu", [(True, True), (False, True), (False, False)])
@pytest.mark.parametrize("momentum,shape", [(0.1, (12, 24, 6, 6)), (None, (6, 5, 3, 3)), (0.3, (9, 130, 7, 7))])
def test_bn_fold_and_act_per_batch_of_a_concatenated_forward(ops, with_res, relu, momentum, shape):
    """Three batches back to back in one tensor: ONE ``pleas_bn_train_fold_batches`` launch folds train-mode BatchNorm on
    every batch's own samples, in order, and ONE ``pleas_bn_act_tracked_batches`` pass applies each batch's map to its
    sample range -- equal to three separate train-mode forwards of the module chain (values, running statistics, counter;
    momentum 0.1 / 0.3 and the cumulative average)."""
    g = torch.Generator().manual_seed(35)
    parts, C = 3, shape[1]
    n = shape[0] // parts
    x = (torch.randn(shape, generator=g) * 2 + 0.5).cuda()
    res = torch.randn(shape, generator=g).cuda()
    bn = torch.nn.BatchNorm2d(C, momentum=momentum).cuda().train()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    ref = copy.deepcopy(bn)
    scales, shifts = ops.BnTrainFold(bn)(x, parts)
    assert tuple(scales.shape) == tuple(shifts.shape) == (parts, C)
    outs = ops.bn_act_tracked(x, scales, shifts, res if with_res else None, relu)
    want_bn = torch.cat([ref(x[i * n:(i + 1) * n]) for i in range(parts)])
    assert torch.allclose(outs[0], want_bn, rtol=1e-4, atol=1e-5)
    last = want_bn + res if with_res else want_bn
    if with_res:
        assert torch.allclose(outs[1], last, rtol=1e-4, atol=1e-5)
    if relu:
        assert torch.allclose(outs[2], torch.relu(last), rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == parts
    # one batch through the same entry points: the old single-map call
    one = ops.BnTrainFold(copy.deepcopy(ref))(x[:n])
    assert one[0].dim() == 1 and torch.allclose(ops.bn_act_tracked(x[:n], one[0], one[1], None, False)[0],
                                               copy.deepcopy(ref)(x[:n]), rtol=1e-4, atol=1e-5)
    with pytest.raises(ops.PleasHipError):
        ops.bn_act_tracked(x[:-1], scales, shifts, None, relu)          # samples do not split into 3 batches


def test_bn_act_rejects_bad_operands(ops):
    x = torch.randn(2, 4, 3, 3).cuda()
    s = torch.ones(4).cuda()
    with pytest.raises(ops.PleasHipError):
        ops.bn_act(x, s[:3], s)
    with pytest.raises(ops.PleasHipError):
        ops.bn_act(x, s, s, torch.randn(2, 4, 3, 2).cuda())
    with pytest.raises(ops.PleasHipError):
        ops.bn_act(x.cpu(), s, s)


@pytest.mark.parametrize("shape,kernel,stride,pad", [((3, 16, 16, 16), (3, 3), 2, 1), ((2, 7, 9, 11), (3, 3), 2, 1),
                                                      ((2, 5, 8, 8), (2, 2), 2, 0), ((1, 4, 7, 7), (3, 2), 1, 1),
                                                      ((2, 3, 5, 5), (5, 5), 3, 2), ((4, 64, 112, 112), (3, 3), 2, 1),
                                                      ((2, 3, 9, 8), (3, 3), 2, 1), ((1, 2, 1, 8), (3, 3), 2, 1)])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_act_maxpool_is_the_pooled_chain(ops, shape, kernel, stride, pad, relu):
    """bn -> relu -> max pooling in one pass == vendor max pooling of the bn_act pass, bit for bit (same fma, and a
    maximum does not round); without scale / shift it is the plain pooling; a NaN tap wins, padding never does."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(33)
    C = shape[1]
    scale, shift = (torch.rand(C, generator=g) - 0.3).cuda(), (torch.randn(C, generator=g) - 2.0).cuda()   # mostly negative
    x = torch.randn(shape, generator=g).cuda()
    x[0, 0, 0, 0] = float("nan")
    x[-1, -1, -1, -1] = float("inf")
    want = F.max_pool2d(ops.bn_act(x, scale, shift, None, relu), kernel, stride, pad)
    got = ops.bn_act_maxpool(x, scale, shift, kernel, stride, pad, relu)
    assert got.shape == want.shape
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(got.nan_to_num(7.0), want.nan_to_num(7.0))
    plain, plain_want = ops.bn_act_maxpool(x, None, None, kernel, stride, pad, False), F.max_pool2d(x, kernel, stride, pad)
    assert torch.equal(plain.nan_to_num(7.0), plain_want.nan_to_num(7.0))


def test_bn_act_maxpool_rejects_bad_operands(ops):
    x = torch.randn(2, 4, 6, 6).cuda()
    s = torch.ones(4).cuda()
    with pytest.raises(ops.PleasHipError):
        ops.bn_act_maxpool(x, s[:3], s, (3, 3), 2, 1)
    with pytest.raises(ops.PleasHipError):
        ops.bn_act_maxpool(x, s, s, (3, 3), 2, 2)          # padding > kernel / 2
    with pytest.raises(ops.PleasHipError):
        ops.bn_act_maxpool(x, s, s, (3, 3), 0, 1)
    with pytest.raises(ops.PleasHipError):
        ops.bn_act_maxpool(x[:, :, :2, :2], s, s, (5, 5), 1, 1)      # window larger than the padded input
    with pytest.raises(ops.PleasHipError):
        ops.bn_act_maxpool(x.cpu(), s, s, (3, 3), 2, 1)


def test_source_forward_pools_the_stem_chain(ops):
    """fuse_bn_act on a ResNet: conv1 -> bn1 -> relu -> maxpool becomes ONE bn_act_maxpool node; the hooked
    convolutions see the same inputs / outputs as in the module-by-module forward."""
    from pleas_merging_amd import resnet as zoo
    from pleas_merging_amd.methods.source_forward import fuse_bn_act, _bn_act_pool

    torch.manual_seed(5)
    model = zoo.resnet18(num_classes=10).cuda().eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
    gm = fuse_bn_act(model)
    pooled = [n for n in gm.graph.nodes if n.op == "call_function" and n.target is _bn_act_pool]
    assert len(pooled) == 1 and pooled[0].args[3:] == ((3, 3), 2, 1, True)
    assert not [n for n in gm.graph.nodes if n.op == "call_module" and n.target == "maxpool"]
    seen = {}
    handles = [m.register_forward_hook(lambda m, i, o, n=n: seen.setdefault(n, []).append((i[0].clone(), o.clone())))
               for n, m in model.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
    x = torch.randn(4, 3, 64, 64, device="cuda")
    with torch.no_grad():
        a, b = model(x), gm(x)
    for h in handles:
        h.remove()
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)
    for name, (first, second) in seen.items():
        assert torch.allclose(first[0], second[0], rtol=1e-4, atol=1e-4), name
        assert torch.allclose(first[1], second[1], rtol=1e-4, atol=1e-4), name


def test_fwd_batch_rejects_kpos_major_with_ragged_channels(ops):
    x = torch.randn(1, 24, 4, 4).cuda()
    w = torch.randn(8, 3, 3, 24).cuda()
    o = torch.randn(1, 8, 4, 4).cuda()
    r = torch.arange(8, dtype=torch.int32, device="cuda")
    batch = ops.FwdBatch(torch.device("cuda"))
    batch.add(x, w, None, o, o, r, r, 8, torch.empty_like(o), 1.0, 1.0, (3, 3), 1, 1, flags=ops.FwdBatch.KPOS_MAJOR)
    with pytest.raises(ops.PleasHipError):
        batch.flush(torch.zeros(1, device="cuda"))


def test_merge_batch_equals_single_tensor_launches(ops):
    """Grouped block merge of many tensors == pleas_merge_blocks per tensor, bit for bit: 16-B and scalar pieces,
    ragged tails, 2-D tensors, absent rows that must not leak (NaN planted in the row an absent source is read from)."""
    from pleas_merging_amd.methods.partial_matching import block_maps

    g = torch.Generator().manual_seed(41)
    cases = [((3, 9, 5, 5), (torch.tensor([0, 2, 4, 6]), torch.tensor([1, 3, 5, 7]), torch.tensor([1, 3, 5, 7, 8]), torch.tensor([0, 2, 4, 6, 8]))),
             ((4, 64, 14, 14), None), ((2, 40, 7, 7), None), ((5, 33), None), ((16, 256, 28, 28), None), ((1, 8, 1, 1), None)]
    batch = ops.MergeBatch(torch.device("cuda"))
    outs, wants = [], []
    for shape, blk in cases:
        C = shape[1]
        if blk is None:
            p = torch.randperm(C, generator=g)
            nm = C // 3
            blk = (torch.arange(nm), p[:nm], torch.arange(nm, C), p[nm:])
        x1, x2 = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
        x2[:, 0] = float("nan") if 0 not in blk[1].tolist() + blk[3].tolist() else x2[:, 0]
        r1, r2, nm = block_maps(blk, "cuda")
        outs.append(batch.add(x1, x2, 1, r1, r2, nm))
        wants.append(ops.merge_blocks(x1, x2, 1, r1, r2, nm))
    batch.flush()
    for o, w in zip(outs, wants):
        assert o.shape == w.shape and torch.equal(o, w)
    # a different tensor list re-plans instead of reusing stale tables
    x1, x2 = torch.randn(2, 6, 4, generator=g).cuda(), torch.randn(2, 6, 4, generator=g).cuda()
    r = torch.arange(6, dtype=torch.int32, device="cuda")
    out = batch.add(x1, x2, 1, r, r, 6)
    batch.flush()
    assert torch.equal(out, (x1 + x2) * 0.5)


def test_merge_batch_subsampled_is_what_a_strided_1x1_layer_reads(ops):
    """``subsample=s``: the grouped merge writes every s-th pixel of every s-th line (bit-equal to slicing the plain merge),
    odd image sizes included -- the input of a 1x1 stride-s convolution as a dense tensor."""
    from pleas_merging_amd.methods.partial_matching import block_maps

    g = torch.Generator().manual_seed(43)
    batch = ops.MergeBatch(torch.device("cuda"))
    outs, wants = [], []
    for shape, s in (((3, 10, 8, 8), 2), ((2, 33, 7, 9), 2), ((4, 64, 28, 28), 2), ((2, 6, 10, 10), 3), ((2, 12, 5, 5), 1)):
        C = shape[1]
        p = torch.randperm(C, generator=g)
        nm = C // 2
        blk = (torch.arange(nm), p[:nm], torch.arange(nm, C), p[nm:])
        x1, x2 = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
        r1, r2, n_merged = block_maps(blk, "cuda")
        outs.append(batch.add(x1, x2, 1, r1, r2, n_merged, subsample=s))
        wants.append(ops.merge_blocks(x1, x2, 1, r1, r2, n_merged)[:, :, ::s, ::s])
    batch.flush()
    for o, w in zip(outs, wants):
        assert o.shape == w.shape and o.is_contiguous() and torch.equal(o, w)
    r = torch.arange(3, dtype=torch.int32, device="cuda")
    with pytest.raises(ops.PleasHipError):      # only [N, C, H, W] tensors merged along the channel axis have lines to skip
        batch.add(torch.zeros(2, 3, 4, device="cuda"), torch.zeros(2, 3, 4, device="cuda"), 1, r, r, 0, subsample=2)


def test_degenerate_sizes_do_not_break_the_grouped_launches(ops):
    """One channel, one pixel, K shorter than a chunk, an all-separate merge, a 1x1 LAP: the smallest inputs every
    grouped launch may see."""
    g = torch.Generator().manual_seed(9)
    # contraction: C = 1 and K = 3 (< one 32-deep chunk), next to an ordinary node
    x1, y1 = torch.randn(3, 1, 1, 1, generator=g).cuda(), torch.randn(3, 1, 1, 1, generator=g).cuda()
    x2, y2 = torch.randn(2, 5, 3, 3, generator=g).cuda(), torch.randn(2, 5, 3, 3, generator=g).cuda()
    m1, m2 = torch.zeros(1, 1, device="cuda"), torch.zeros(5, 5, device="cuda")
    batch = ops.GramBatch([m1, m2], ops.EPI_INNER)
    batch.add(x1, y1, 1, 0)
    batch.add(x2, y2, 1, 1)
    batch.flush(accumulate=False)
    assert torch.allclose(m1, (x1.flatten() * y1.flatten()).sum().view(1, 1), atol=1e-6)
    want = torch.einsum("nchw,ndhw->cd", x2.double(), y2.double()).float()
    assert torch.allclose(m2, want, atol=1e-5)
    # merge: no merged rows at all (ratio 1), single-element rows
    a, b = torch.randn(2, 3, 1, 1, generator=g).cuda(), torch.randn(2, 3, 1, 1, generator=g).cuda()
    r1 = torch.tensor([0, 1, 2, -1, -1, -1], dtype=torch.int32, device="cuda")
    r2 = torch.tensor([-1, -1, -1, 0, 1, 2], dtype=torch.int32, device="cuda")
    mb = ops.MergeBatch(torch.device("cuda"))
    out = mb.add(a, b, 1, r1, r2, 0)
    mb.flush()
    assert torch.equal(out, torch.cat([a, b], 1))
    # LAP: n = 1
    assert ops.solve_lsa_batched([torch.tensor([[3.0]]).cuda()])[0].tolist() == [0]


def test_masked_adam_matches_torch(ops):
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(1000, generator=g)
    mask = (torch.rand(1000, generator=g) > 0.2).float()
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=5e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 10)
    p, m, v = p0.clone().cuda(), torch.zeros(1000).cuda(), torch.zeros(1000).cuda()
    for step in range(1, 12):
        grad = torch.randn(1000, generator=g) * 10 ** float(torch.randint(-6, 1, (1,), generator=g))
        p_ref.grad = grad * mask
        lr = opt.param_groups[0]["lr"]
        opt.step()
        sched.step()
        ops.masked_adam(p, grad.cuda(), mask.cuda(), m, v, lr, step)
    assert _rel(p.cpu(), p_ref.detach()) < 1e-6
    assert torch.equal(p.cpu()[mask == 0], p0[mask == 0])


def test_sqerr(ops):
    a, b = torch.randn(100003), torch.randn(100003)
    out = torch.zeros(1, device="cuda")
    diff = torch.empty(100003, device="cuda")
    ops.sqerr(a.cuda(), b.cuda(), 1.0 / a.numel(), out, diff=diff, dscale=2.0 / a.numel())
    assert abs(float(out) - float(((a - b) ** 2).mean())) < 1e-5
    assert torch.allclose(diff.cpu(), 2 * (a - b) / a.numel(), rtol=1e-6, atol=1e-12)
    ops.sqerr(a.cuda(), b.cuda(), 1.0 / a.numel(), out, accumulate=True)
    assert abs(float(out) - 2 * float(((a - b) ** 2).mean())) < 2e-5


# ------------------------------------------------------------------------------------------ grouped gram launch
def test_gram_batch_matches_per_node_sums(ops):
    """One grouped launch over nodes of mixed shapes/groups == sum of single-node launches."""
    g = torch.Generator().manual_seed(11)
    shapes = [((4, 8, 6, 6), 0), ((4, 8, 3, 3), 0), ((4, 130, 9, 9), 1), ((4, 130, 5, 5), 1), ((4, 64, 28, 28), 2),
              ((4, 256, 14, 14), 3), ((4, 256, 14, 14), 3), ((4, 256), 3), ((2, 64, 112, 112), 2)]
    sizes = {0: 8, 1: 130, 2: 64, 3: 256}
    for epi in (ops.EPI_NEG_CDIST, ops.EPI_INNER):
        mats = [torch.full((sizes[k], sizes[k]), 0.25, device="cuda") for k in range(4)]
        want = [m.clone() for m in mats]
        batch = ops.GramBatch(mats, epi)
        for rep in range(2):  # second round reuses the cached plan with new operand pointers
            for shp, grp in shapes:
                x = torch.randn(shp, generator=g).cuda()
                y = (0.5 * x.cpu() + torch.randn(shp, generator=g)).cuda()
                batch.add(x, y, 1, grp)
                ops.gram_accum(x, y, 1, want[grp], epi, accumulate=True)
            batch.flush(accumulate=True)
        for got, ref in zip(mats, want):
            assert _rel(got, ref) < 1e-5, _rel(got, ref)  # different K splits: only rounding may differ


@pytest.mark.parametrize("shape", [(16, 2048, 7, 7), (3, 136, 7, 7), (2, 70, 5, 3), (1, 33, 3, 3), (5, 130, 9, 9)])
def test_gram_batch_padded_pixel_form_vs_fp64(ops, shape):
    """Images with HW % 4 != 0 in the grouped launch: the K axis of (sample, pixel padded to a multiple of four) with under-aligned
    16-byte loads (round 5; the single-node entry point keeps one pixel per load) against fp64, both epilogues, and the last
    row's clamped run (the tensor ends inside a padded group)."""
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    y = 0.6 * x + 0.5 * torch.randn(shape, generator=g)
    C = shape[1]
    for epi, want in ((ops.EPI_INNER, orc.cross_features_inner_product(x.double(), y.double(), 1)),
                      (ops.EPI_NEG_CDIST, orc.cross_features_cdist_f64(x, y, 1))):
        mats = [torch.full((C, C), float("nan"), device="cuda")]
        batch = ops.GramBatch(mats, epi)
        batch.add(x.cuda(), y.cuda(), 1, 0)
        batch.flush(accumulate=False)
        assert _rel(mats[0].cpu(), want) < 2e-6, (shape, epi, _rel(mats[0].cpu(), want))


@pytest.mark.parametrize("epi", ["cdist", "inner"])
def test_gram_batch_derived_affine_nodes_match_contracted_ones(ops, epi):
    """A node declared as a per-channel affine image of another node (eval-mode BatchNorm of a convolution output) is
    derived from the source's products, norms and row sums; it must agree with contracting the transformed tensors."""
    g = torch.Generator().manual_seed(77)
    epilogue = ops.EPI_NEG_CDIST if epi == "cdist" else ops.EPI_INNER
    shapes = [(4, 96, 7, 7), (3, 40, 6, 6), (2, 130, 1, 1)]      # scalar / vector pieces, ragged tiles, linear nodes
    mats_a = [torch.zeros(s[1], s[1], device="cuda") for s in shapes]
    mats_b = [torch.zeros(s[1], s[1], device="cuda") for s in shapes]
    direct, derived = ops.GramBatch(mats_a, epilogue), ops.GramBatch(mats_b, epilogue)
    for gi, shape in enumerate(shapes):
        C = shape[1]
        x = (torch.randn(shape, generator=g) * 1.5 + 0.7).cuda()      # non-zero mean: the shift terms matter
        y = (torch.randn(shape, generator=g) * 0.8 - 0.4).cuda()
        sx, tx = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
        sy, ty = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
        view = (1, C) + (1,) * (len(shape) - 2)
        direct.add(x, y, 1, gi)
        direct.add(x * sx.view(view) + tx.view(view), y * sy.view(view) + ty.view(view), 1, gi)
        src = derived.add(x, y, 1, gi)
        derived.add_derived(src, sx, tx, sy, ty, gi)
    direct.flush(accumulate=False)
    derived.flush(accumulate=False)
    for a, b in zip(mats_a, mats_b):
        assert _rel(b, a) < 2e-5
    with pytest.raises(ops.PleasHipError):
        derived.add_derived(0, sx, tx, sy, ty, 0)      # nothing queued yet after the flush


@pytest.mark.parametrize("shape", [(16, 1024, 14, 14), (16, 2048, 7, 7)])
def test_gram_batch_derived_train_mode_batchnorm_at_full_width(ops, shape):
    """The reduce pass's derived branch at the widths of ResNet-101's layer3 / layer4 residual streams, fed the way the
    drivers' mode feeds it (no .eval() before activation_matching, run_domainnet.py:172-186): the per-batch statistics of
    a TRAIN-mode BatchNorm2d folded on the device (``bn_train_fold``) are the affine map of the derived node.  Against
    fp64: -cdist of the two train-mode BatchNorm outputs (reference activation_matching.py:31-46), and against the same
    launch contracting those outputs."""
    import torch.nn.functional as F

    B, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    x = torch.relu(torch.randn(shape, generator=g) * 1.3 + 0.4) * (0.5 + torch.rand(1, C, 1, 1, generator=g))
    y = torch.relu(torch.randn(shape, generator=g) * 0.9 + 0.2) * (0.5 + torch.rand(1, C, 1, 1, generator=g))
    bns = []
    for _ in range(2):
        bn = torch.nn.BatchNorm2d(C).train()
        with torch.no_grad():
            bn.weight.copy_(0.5 + torch.rand(C, generator=g))
            bn.bias.copy_(0.3 * torch.randn(C, generator=g))
        bns.append(bn)
    want = []
    for t, bn in zip((x, y), bns):
        want.append(F.batch_norm(t.double(), None, None, bn.weight.double(), bn.bias.double(), True, 0.0, bn.eps))
    a = want[0].movedim(1, 0).reshape(C, -1)
    b = want[1].movedim(1, 0).reshape(C, -1)
    ref = -torch.cdist(a, b)
    xg, yg = x.cuda(), y.cuda()
    gbn = [copy_bn.cuda() for copy_bn in bns]
    (sx, tx), (sy, ty) = ops.bn_train_fold(gbn[0], xg), ops.bn_train_fold(gbn[1], yg)
    mats = [torch.zeros(C, C, device="cuda") for _ in range(3)]
    batch = ops.GramBatch(mats, ops.EPI_NEG_CDIST)
    src = batch.add(xg, yg, 1, 0)
    batch.add_derived(src, sx, tx, sy, ty, 1)                                      # derived in the reduce pass
    view = (1, C, 1, 1)
    batch.add(xg * sx.view(view) + tx.view(view), yg * sy.view(view) + ty.view(view), 1, 2)   # the same tensors, contracted
    batch.flush(accumulate=False)
    derived, contracted = _rel(mats[1].cpu(), ref), _rel(mats[2].cpu(), ref)
    print("C = %d: derived %.2e, contracted %.2e vs fp64" % (C, derived, contracted))
    assert derived < max(3 * contracted, 2e-6), (derived, contracted)
    assert int(gbn[0].num_batches_tracked) == 1


def test_gram_batch_deterministic_and_overwrite(ops):
    x, y = torch.randn(8, 256, 14, 14).cuda(), torch.randn(8, 256, 14, 14).cuda()
    outs = []
    for _ in range(2):
        m = torch.full((256, 256), 7.0, device="cuda")
        b = ops.GramBatch([m], ops.EPI_NEG_CDIST)
        b.add(x, y, 1, 0)
        b.add(y, x, 1, 0)
        b.flush(accumulate=False)
        outs.append(m.clone())
    assert torch.equal(outs[0], outs[1])
    ref = ops.cross_features_cdist(x, y, 1) + ops.cross_features_cdist(y, x, 1)
    assert torch.allclose(outs[0], ref, rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------ PLeaS layer kernels
WGRAD_CASES = [
    # N, Cout, Cin, H, W, k, stride, pad
    (4, 256, 64, 14, 14, 1, 1, 0),     # 1x1: direct loader, TN = 64
    (4, 64, 256, 14, 14, 1, 1, 0),     # TM = 64
    (3, 96, 80, 7, 7, 1, 1, 0),        # HW = 49: scalar loads, ragged tiles
    (4, 128, 128, 14, 14, 3, 1, 1),    # 3x3 shifted loader, padding
    (2, 40, 24, 9, 11, 3, 1, 1),       # ragged everything, non-square image
    (4, 128, 64, 28, 28, 3, 2, 1),     # 3x3 stride 2
    (4, 256, 128, 28, 28, 1, 2, 0),    # 1x1 stride 2 (downsample)
    (2, 64, 64, 56, 56, 3, 1, 1),      # long pixel axis -> split into slabs + reduce
    (16, 64, 32, 1, 1, 1, 1, 0),       # linear-like (HW = 1)
    # fewer than 16 input channels: rows of the tile are (channel, tap) pairs ("virtual channels")
    (4, 64, 3, 224, 224, 7, 2, 3),     # the ResNet stem (K = 147, slabs + reduce)
    (2, 8, 3, 9, 11, 3, 1, 1),         # tiny, ragged, one tile
    (2, 20, 5, 12, 12, 5, 2, 2),       # 5x5 stride 2: 125 virtual channels, HWo = 36
    (3, 70, 15, 7, 7, 3, 1, 1),        # 135 virtual channels: TN = 128, HW = 49 (scalar residual loads)
    # images with HW % 4 != 0 on stride-1 same-size layers (one pixel per load; a padded-pixel 16-byte form was built and measured
    # in round 5: no gain on these short-K layers, profiles/r05_padk_ab.txt -- the cases stay)
    (4, 96, 64, 7, 7, 3, 1, 1),        # 3x3 at 7 x 7 (layer4)
    (16, 512, 512, 7, 7, 3, 1, 1),     # the same at ResNet size: 128 x 128 tiles, 25 chunks
    (16, 2048, 512, 7, 7, 1, 1, 0),    # 1x1 at 7 x 7, batch 16 (layer4 conv3)
    (2, 20, 32, 5, 5, 5, 1, 2),        # 5x5 "same" on a 5 x 5 image (HW = 25)
    (1, 33, 17, 3, 3, 1, 1, 0),        # one sample, HW = 9: the last row's run is clamped at the tensor's end
]


@pytest.mark.parametrize("N,Cout,Cin,H,W,k,stride,pad", WGRAD_CASES)
def test_wgrad_batch_matches_torch(ops, N, Cout, Cin, H, W, k, stride, pad):
    g = torch.Generator().manual_seed(N * Cout + Cin + k)
    ip = torch.randn(N, Cin, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    resid = torch.randn(N, Cout, Ho, Wo, generator=g)
    w = torch.zeros(Cout, Cin, k, k)
    want = torch.ops.aten.convolution_backward(resid.double(), ip.double(), w.double(), None, [stride, stride], [pad, pad],
                                               [1, 1], False, [0, 0], 1, [False, True, False])[1]
    batch = ops.WgradBatch(torch.device("cuda"))
    grad = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
    grad2 = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
    batch.add(resid.cuda(), ip.cuda(), grad, (k, k), stride, pad)
    batch.add((2 * resid).cuda(), ip.cuda(), grad2, (k, k), stride, pad)   # two layers in one launch
    batch.flush()
    assert _rel(grad.cpu(), want) < 3e-6
    assert _rel(grad2.cpu(), 2 * want) < 3e-6


def test_wgrad_linear(ops):
    g = torch.Generator().manual_seed(1)
    ip, resid = torch.randn(16, 300, generator=g), torch.randn(16, 70, generator=g)
    grad = torch.empty(70, 300, device="cuda")
    batch = ops.WgradBatch(torch.device("cuda"))
    batch.add(resid.cuda(), ip.cuda(), grad)
    batch.flush()
    assert _rel(grad.cpu(), resid.double().t() @ ip.double()) < 2e-6


def test_target_residual_and_loss(ops):
    g = torch.Generator().manual_seed(2)
    o1, o2 = torch.randn(3, 9, 5, 5, generator=g), torch.randn(3, 9, 5, 5, generator=g)
    b = (torch.tensor([0, 2, 4, 6]), torch.tensor([1, 3, 5, 7]), torch.tensor([1, 3, 5, 7, 8]),
         torch.tensor([0, 2, 4, 6, 8]))
    from pleas_merging_amd.methods.partial_matching import block_maps

    r1, r2, nm = block_maps(b, "cuda")
    target = torch.cat([(o1[:, b[0]] + o2[:, b[1]]) / 2, o1[:, b[2]], o2[:, b[3]]], 1)
    out = torch.randn(target.shape, generator=g)
    n = out.numel()
    parts = torch.zeros(2, ops.target_residual_max_partials(), device="cuda")
    buf = out.clone().cuda()
    cnt = ops.target_residual(buf, o1.cuda(), o2.cuda(), r1, r2, nm, 2.0 / n, parts[1])
    assert torch.allclose(buf.cpu(), 2 * (out - target) / n, rtol=1e-6, atol=1e-9)
    loss = torch.zeros(2, device="cuda")
    ops.loss_final(parts, torch.tensor([0, cnt], dtype=torch.int32, device="cuda"),
                   torch.tensor([1.0, 1.0 / n], device="cuda"), loss)
    assert float(loss[0]) == 0.0 and abs(float(loss[1]) - float(((out - target) ** 2).mean())) < 1e-5


# ------------------------------------------------------------------------------------------ normal equations
NEQ_CASES = [
    # N, Cin, H, W, k, stride, pad
    (4, 64, 14, 14, 1, 1, 0),      # direct loader, one 64-tile
    (4, 200, 14, 14, 1, 1, 0),     # 2x2 block tiles, ragged
    (3, 40, 7, 7, 1, 1, 0),        # HW = 49 scalar loads
    (4, 48, 10, 10, 3, 1, 1),      # 3x3 stride 1: lag classes (29 of 45 blocks contracted), 64-tile, 16-byte loads
    (2, 130, 12, 12, 3, 2, 1),     # stride 2 (shifted loader, every block contracted), two tiles per position
    (2, 64, 56, 56, 1, 1, 0),      # long pixel axis -> slabs + reduce
    (16, 96, 1, 1, 1, 1, 0),       # linear-like
    (2, 256, 14, 14, 3, 1, 1),     # K = 2304 (ResNet-101 layer3 conv2): lag classes, 128-tiles, W = 14 -> shifts 0..3
    (2, 512, 7, 7, 3, 1, 1),       # K = 4608 (layer4 conv2): lag classes with scalar loads (HW = 49)
    (3, 200, 6, 10, 3, 1, 1),      # ragged 128-tiles, H != W
    (2, 32, 56, 56, 3, 1, 1),      # lag classes through slabs + reduce (196 chunks of the pixel axis)
    (5, 24, 8, 8, 5, 1, 2),        # 5x5 "same": 157 of 325 blocks contracted
    (2, 20, 2, 3, 3, 1, 1),        # image smaller than the kernel: empty windows
]


@pytest.mark.parametrize("N,Cin,H,W,k,stride,pad", NEQ_CASES)
def test_normal_eq_accum_matches_unfold(ops, N, Cin, H, W, k, stride, pad):
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(N + Cin + k)
    K = k * k * Cin
    A = torch.zeros(K, K, device="cuda")
    batch = ops.NormalEqBatch(torch.device("cuda"))
    want = torch.zeros(K, K, dtype=torch.float64)
    for rep in range(2):  # accumulates over batches
        ip = torch.randn(N, Cin, H, W, generator=g)
        U = F.unfold(ip.double(), k, 1, pad, stride)                                   # N, (ci, r), L
        U = U.view(N, Cin, k * k, -1).permute(0, 3, 2, 1).reshape(-1, K)               # rows x (r, ci)
        want += U.t() @ U
        batch.add(ip.cuda(), A, (k, k), stride, pad)
        batch.flush()
    info = batch.plan_info()
    lag = k > 1 and stride == 1 and 2 * pad == k - 1
    assert info["blocks_to_finalize"] == ({3: 16, 5: 168}[k] if lag else 0)
    batch.finalize()                  # copies / transposes of the contracted lag-class blocks (once, after the last batch)
    batch.finalize()                  # idempotent
    got = A.cpu().double()
    T = 128 if Cin > 64 else 64
    tiles = -(-Cin // T)
    blk = lambda idx: (idx // Cin) * tiles + (idx % Cin) // T                      # block-tile index of a row/col
    rows = torch.arange(K)
    lower = blk(rows)[:, None] >= blk(rows)[None, :]
    assert _rel(got[lower], want[lower]) < 3e-6
    assert (got[~lower] == 0).all()                                                    # strict upper tiles untouched


def test_wgrad_accumulate_kpos_major(ops):
    g = torch.Generator().manual_seed(9)
    ip, resid = torch.randn(4, 40, 9, 9, generator=g), torch.randn(4, 24, 9, 9, generator=g)
    w = torch.zeros(24, 40, 3, 3)
    want = torch.ops.aten.convolution_backward(resid.double(), ip.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False,
                                               [0, 0], 1, [False, True, False])[1]
    out = torch.ones(24, 9 * 40, device="cuda")
    batch = ops.WgradBatch(torch.device("cuda"))
    for _ in range(2):
        batch.add(resid.cuda(), ip.cuda(), out, (3, 3), 1, 1, flags=ops.WgradBatch.ACCUMULATE | ops.WgradBatch.KPOS_MAJOR)
        batch.flush()
    ref = 1 + 2 * want.reshape(24, 40, 9).permute(0, 2, 1).reshape(24, 360)
    assert _rel(out.cpu(), ref) < 3e-6


# ------------------------------------------------------------------------------------------ batched SPD solve
def test_cholesky_solve_batched_mixed_sizes(ops):
    g = torch.Generator().manual_seed(21)
    sizes = [(1, 3), (5, 1), (64, 7), (65, 100), (130, 17), (200, 256), (513, 40), (1000, 129)]
    As, Bts, refs = [], [], []
    for K, N in sizes:
        M = torch.randn(K, K + 8, generator=g, dtype=torch.float64)
        A = M @ M.t() / (K + 8) + 0.5 * torch.eye(K, dtype=torch.float64)
        Bt = torch.randn(N, K, generator=g, dtype=torch.float64)
        refs.append(torch.linalg.solve(A, Bt.t()).t())
        As.append(torch.tril(A).float().cuda().contiguous())       # only the lower triangle is read
        Bts.append(Bt.float().cuda().contiguous())
    info = ops.cholesky_solve_batched(As, Bts, ridge=0.0)
    assert (info.cpu() == 0).all()
    for (K, N), got, want in zip(sizes, Bts, refs):
        assert _rel(got.cpu(), want) < 2e-5, (K, N, _rel(got.cpu(), want))


def test_cholesky_solve_ridge_and_breakdown_flag(ops):
    g = torch.Generator().manual_seed(22)
    K = 96
    M = torch.randn(K, 10, generator=g, dtype=torch.float64)
    A = M @ M.t()                                                    # rank 10: singular
    Bt = torch.randn(4, K, generator=g, dtype=torch.float64)
    lam = 1e-3
    want = torch.linalg.solve(A + lam * A.diagonal().mean() * torch.eye(K, dtype=torch.float64), Bt.t()).t()
    a, b = A.float().cuda().contiguous(), Bt.float().cuda().contiguous()
    info = ops.cholesky_solve_batched([a], [b], ridge=lam)
    assert int(info[0]) == 0 and _rel(b.cpu(), want) < 5e-3
    a2 = (-torch.eye(8)).cuda().contiguous()                         # not positive definite -> flagged, no NaN trap
    info = ops.cholesky_solve_batched([a2], [torch.ones(2, 8).cuda()], ridge=0.0)
    assert int(info[0]) == 1


# ------------------------------------------------------------------------------------------ fused forward + target + residual + loss
FWD_CASES = [
    # N, Cout, Cin, H, W, k, stride, pad, bias
    (4, 256, 64, 14, 14, 1, 1, 0, False),
    (4, 64, 256, 14, 14, 1, 1, 0, False),     # TM = 64
    (3, 96, 80, 7, 7, 1, 1, 0, False),        # ragged pixels / channels
    (4, 128, 128, 14, 14, 3, 1, 1, False),    # 3x3 with padding
    (2, 40, 24, 9, 11, 3, 1, 1, False),       # non-square image
    (4, 128, 64, 28, 28, 3, 2, 1, False),     # stride 2
    (3, 72, 96, 10, 7, 3, 1, 1, False),       # kernel-position-major with ragged rows / pixels, non-square image
    (2, 64, 32, 12, 12, 5, 1, 2, False),      # 5x5 taps
    (2, 64, 3, 32, 32, 7, 2, 3, False),       # stem geometry: Kd = 147 (scalar weight loads)
    (16, 70, 300, 1, 1, 1, 1, 0, True),       # linear layer with bias
    # flat-shift tile forms (stride 1, "same" padding, Cin % 32 == 0): one LDS image per channel block, taps = shifts
    (5, 200, 96, 14, 14, 1, 1, 0, False),     # 1x1, 16-B pixel loads, ragged last pixel tile (980 pixels) and channel tile
    (3, 136, 64, 7, 7, 1, 1, 0, False),       # 1x1, HW = 49: scalar pixel loads, tiles straddle samples
    (16, 40, 64, 1, 1, 1, 1, 0, True),        # linear layer on the flat path (HW = 1), TM = 64, bias
    (5, 136, 64, 14, 14, 3, 1, 1, False),     # 3x3: tiles straddle samples and rows, every border case
    (2, 64, 32, 56, 56, 3, 1, 1, False),      # 3x3 at W = 56: widest halo (242 data columns), TM = 64
    (3, 130, 96, 7, 7, 3, 1, 1, False),       # 3x3 at 7x7: halo 8, three samples per tile
    (2, 72, 32, 9, 11, 5, 1, 2, False),       # 5x5 "same", non-square image
    (3, 96, 64, 28, 28, 3, 1, 1, True),       # 3x3 at W = 28 with bias
]


@pytest.mark.parametrize("N,Cout,Cin,H,W,k,stride,pad,bias", FWD_CASES)
def test_fwd_batch_matches_conv_and_target(ops, N, Cout, Cin, H, W, k, stride, pad, bias):
    import torch.nn.functional as F
    from pleas_merging_amd.methods.partial_matching import block_maps

    g = torch.Generator().manual_seed(Cout * 7 + Cin + k)
    ip = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    nm, ns = Cout - 2 * (Cout // 5), Cout // 5          # merged / separate output units: Csrc = nm + ns
    Csrc = nm + ns
    pm = torch.randperm(Csrc, generator=g)
    blk = (torch.arange(nm), pm[:nm], torch.arange(nm, Csrc), pm[nm:])
    o1, o2 = torch.randn(N, Csrc, Ho, Wo, generator=g), torch.randn(N, Csrc, Ho, Wo, generator=g)
    target = torch.cat([(o1[:, blk[0]] + o2[:, blk[1]]) / 2, o1[:, blk[2]], o2[:, blk[3]]], 1)
    out = F.conv2d(ip.double(), w.double(), b.double() if bias else None, stride, pad)
    numel = out.numel()
    want = 2 * (out - target.double()) / numel
    want_loss = float(((out - target.double()) ** 2).mean())
    r1, r2, nmerged = block_maps(blk, "cuda")
    batch = ops.FwdBatch(torch.device("cuda"))
    resid = [torch.full((N, Cout, Ho, Wo), float("nan"), device="cuda") for _ in range(2)]
    loss = torch.zeros(2, device="cuda")
    wd = w.cuda().contiguous()
    # second layer of the launch: the same weights kernel-position-major [Cout][KH][KW][Cin] where that layout applies
    kpos = Cin % 32 == 0 and k > 1
    wk = w.permute(0, 2, 3, 1).contiguous().cuda() if kpos else wd
    for i in range(2):   # two layers in one launch
        batch.add(ip.cuda(), wk if i else wd, b.cuda() if bias else None, o1.cuda(), o2.cuda(), r1, r2, nmerged, resid[i],
                  2.0 / numel, 1.0 / numel, (k, k), stride, pad, flags=ops.FwdBatch.KPOS_MAJOR if (i and kpos) else 0)
    batch.flush(loss)
    for i in range(2):
        assert _rel(resid[i].cpu(), want) < 5e-6
        assert abs(float(loss[i]) - want_loss) < 1e-5 * max(1.0, want_loss)


# ------------------------------------------------------------------------------------------ train-mode BatchNorm fold
@pytest.mark.parametrize("shape", [(4, 8, 5, 5), (16, 64, 56, 56), (2, 2048, 7, 7), (3, 37, 1, 9), (8, 256, 14, 14)])
@pytest.mark.parametrize("momentum", [0.1, None])
def test_bn_train_fold_equals_train_mode_batchnorm(shape, momentum):
    """``pleas_bn_train_fold`` + ``pleas_bn_act``: the values, the running statistics and the batch counter of a
    train-mode ``BatchNorm2d`` forward (torch, fp64 on the CPU as the truth), over two consecutive batches."""
    from pleas_merging_amd import hip_ops

    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C, momentum=momentum)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(C, generator=g))
        bn.bias.copy_(0.2 * torch.randn(C, generator=g))
    ref = copy.deepcopy(bn).double().train()
    dev = copy.deepcopy(bn).cuda().train()
    for step in range(2):
        x = 3.0 + 2.0 * torch.randn(shape, generator=g)        # mean >> 0: E[x^2] - mean^2 must not cancel in fp32
        want = ref(x.double())
        scale, shift = hip_ops.bn_train_fold(dev, x.cuda())
        got = hip_ops.bn_act(x.cuda(), scale, shift, None, relu=False)
        assert _rel(got.cpu(), want) < 1e-6, (step, _rel(got.cpu(), want))
        assert _rel(dev.running_mean.cpu(), ref.running_mean) < 1e-6 and _rel(dev.running_var.cpu(), ref.running_var) < 1e-6
        assert int(dev.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
    # no running statistics at all: batch statistics, nothing to update
    free = torch.nn.BatchNorm2d(C, track_running_stats=False).cuda()
    x = torch.randn(shape, generator=g)
    scale, shift = hip_ops.bn_train_fold(free, x.cuda())
    want = torch.nn.functional.batch_norm(x.double(), None, None, free.weight.double().cpu(), free.bias.double().cpu(), True)
    assert _rel(hip_ops.bn_act(x.cuda(), scale, shift, None, relu=False).cpu(), want) < 1e-6
    with pytest.raises(ValueError):
        hip_ops.bn_train_fold(dev, torch.randn(1, C, 1, 1).cuda())


def test_fwd_batch_flat_forms_random_geometries(ops):
    """Property test of the flat-shift tile forms over random stride-1 "same" geometries (image sizes from 1x1 to 30x30,
    tiles that straddle up to 128 samples, ragged channel tiles, 1x1 / 3x3 / 5x5, with and without bias): every layer of
    ONE grouped launch against fp64 convolution + block-merged target, forms mixed in the launch as in a real update."""
    import torch.nn.functional as F
    from pleas_merging_amd.methods.partial_matching import block_maps

    g = torch.Generator().manual_seed(1234)
    rnd = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    batch = ops.FwdBatch(torch.device("cuda"))
    cases = []
    for i in range(24):
        k = (1, 3, 5, 1, 3, 3)[i % 6]
        pad = k // 2
        H, W = (1, 1) if i == 3 else (rnd(2, 30), rnd(2, 30))
        if k == 5:
            H, W = min(H, 12), min(W, 12)           # halo 2 * (W + 1) must fit the image rows of the k x k form
        N, Cin, Cout = rnd(1, 9), 32 * rnd(1, 4), rnd(8, 200)
        ip = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b = torch.randn(Cout, generator=g) if i % 4 == 1 else None
        ns = Cout // 4
        nm = Cout - 2 * ns
        Csrc = nm + ns
        pm = torch.randperm(Csrc, generator=g)
        blk = (torch.arange(nm), pm[:nm], torch.arange(nm, Csrc), pm[nm:])
        o1, o2 = torch.randn(N, Csrc, H, W, generator=g), torch.randn(N, Csrc, H, W, generator=g)
        target = torch.cat([(o1[:, blk[0]] + o2[:, blk[1]]) / 2, o1[:, blk[2]], o2[:, blk[3]]], 1)
        out = F.conv2d(ip.double(), w.double(), b.double() if b is not None else None, 1, pad)
        numel = out.numel()
        r1, r2, nmerged = block_maps(blk, "cuda")
        resid = torch.full((N, Cout, H, W), float("nan"), device="cuda")
        wk = w.permute(0, 2, 3, 1).contiguous().cuda() if k > 1 else w.cuda().contiguous()
        keep = (ip.cuda(), wk, b.cuda() if b is not None else None, o1.cuda(), o2.cuda(), r1, r2, resid)
        batch.add(keep[0], wk, keep[2], keep[3], keep[4], r1, r2, nmerged, resid, 2.0 / numel, 1.0 / numel, (k, k), 1, pad,
                  flags=ops.FwdBatch.KPOS_MAJOR if k > 1 else 0)
        cases.append((keep, 2 * (out - target.double()) / numel, float(((out - target.double()) ** 2).mean()), (N, Cout, Cin, H, W, k)))
    loss = torch.zeros(len(cases), device="cuda")
    batch.flush(loss)
    for i, (keep, want, want_loss, geo) in enumerate(cases):
        assert _rel(keep[7].cpu(), want) < 5e-6, (geo, _rel(keep[7].cpu(), want))
        assert abs(float(loss[i]) - want_loss) < 1e-5 * max(1.0, want_loss), geo


def test_channel_sum_is_the_bias_gradient(ops):
    g = torch.Generator().manual_seed(3)
    for shape in [(16, 1000), (4, 70, 14, 14), (3, 9, 5, 7), (1, 1, 1, 1), (300, 5), (2, 64, 112, 112), (3, 17, 33, 4)]:
        x = torch.randn(shape, generator=g)
        out = torch.full((shape[1],), float("nan"), device="cuda")
        ops.channel_sum(x.cuda(), out)
        want = x.double().sum(dim=[d for d in range(x.dim()) if d != 1])
        assert _rel(out.cpu(), want) < 2e-6, shape
        again = torch.empty_like(out)
        ops.channel_sum(x.cuda(), again)
        assert torch.equal(out, again)      # deterministic


# ------------------------------------------------------------------------------------------ exchange step through the C-ABI
_ALLREDUCE_CHILD = r"""
import ctypes, sys, time
sys.path.insert(0, %r)
import torch
from pleas_merging_amd import _lib

class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]

torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
rccl = None
for name in ("librccl.so.1", "/opt/rocm/lib/librccl.so.1"):
    try:
        rccl = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)      # the instance pleas_allreduce_sum will find in the process
        break
    except OSError:
        pass
assert rccl is not None, "no RCCL on this box"
uid, comm = UniqueId(), ctypes.c_void_p()
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0      # a ONE-rank communicator: all one GPU allows
lib = _lib.lib()
n = 10317824                                                            # ResNet-101's cost arena: 71 group matrices, 41 MB
buf = torch.randn(n, device="cuda")
keep = buf.clone()
stream = torch.cuda.current_stream().cuda_stream
assert lib.pleas_allreduce_sum(buf.data_ptr(), n, comm, stream) == 0, lib.pleas_last_error()
torch.cuda.synchronize()
assert torch.equal(buf, keep)                                           # a sum over one rank is the identity
t0 = time.perf_counter()
for _ in range(10):
    assert lib.pleas_allreduce_sum(buf.data_ptr(), n, comm, stream) == 0
torch.cuda.synchronize()
print("ALLREDUCE_OK %%.1f us per call" %% ((time.perf_counter() - t0) / 10 * 1e6))
assert lib.pleas_allreduce_sum(buf.data_ptr(), 0, comm, stream) == 0 and lib.pleas_allreduce_sum(None, 4, comm, stream) == -22
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclCommDestroy(comm)
"""


def test_allreduce_sum_entry_point_on_a_one_rank_rccl_communicator():
    """``pleas_allreduce_sum(buf, n, ncclComm_t, stream)`` -- the exchange step for consumers of the
    C-ABI that are not under torch.distributed.  A one-rank RCCL communicator made with RCCL's own API (ncclGetUniqueId /
    ncclCommInitRank), the 41 MB cost arena, in place; own process, so that a stuck communicator cannot hang the suite."""
    import subprocess
    import sys

    from conftest import REPO

    out = subprocess.run([sys.executable, "-c", _ALLREDUCE_CHILD % REPO], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALLREDUCE_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    print(out.stdout.strip().splitlines()[-1])
