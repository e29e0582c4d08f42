"""The SAME job returns the SAME bits (missing 2).

The reference's solver is deterministic given its costs (pleas/core/solvers.py:18-33), and so are its loops given deterministic
operators; "bit-exact for integer work" means nothing if the integer output changes between two runs on one box.  Rounds 1-4
ran every convolution of the frozen source / twin forwards on the vendor's kernels, whose 3 x 3 picks split K with atomics
(tools/r05/probe_conv_classes.py): costs moved by 1e-5 between two runs and near-tie groups flipped.  Since round 5 those
layers -- and, because that is also the fastest arrangement in the job, the 1 x 1 layers -- run on ``pleas_conv2d_fwd``
(methods/source_forward.py: SOURCE_CONV = "all"), every own kernel reduces in a fixed
order, and the library's DEFAULT path is held here to ``torch.equal`` on costs, assignments and trained weights.
"""
import os

import pytest
import torch
import torch.nn.functional as F

import test_hip_fullsize_dp as dp

pytestmark = pytest.mark.gpu


GEOMETRIES = [
    # (N, Cin, H, W, Cout, k, stride, pad, bias)
    (4, 64, 56, 56, 64, 3, 1, 1, False),       # flat-shift k x k form, kernel-position-major weights
    (3, 128, 28, 28, 96, 3, 1, 1, True),       # Cout not a multiple of the tile, bias
    (2, 128, 56, 56, 128, 3, 2, 1, False),     # strided 3 x 3: general form
    (2, 3, 224, 224, 64, 7, 2, 3, False),      # the stem: K = 147, scalar weight loads
    (5, 512, 7, 7, 512, 3, 1, 1, False),       # 7 x 7 images (HW % 4 != 0)
    (2, 32, 17, 23, 40, 5, 1, 2, True),        # H != W, 5 x 5
    (2, 48, 15, 15, 24, 3, 1, 0, False),       # "valid" padding, Cin % 32 != 0
    (3, 256, 14, 14, 1024, 1, 1, 0, False),    # 1 x 1 (mode "all")
    (2, 256, 56, 56, 512, 1, 2, 0, False),     # strided 1 x 1 (mode "all")
]


@pytest.mark.parametrize("geo", GEOMETRIES)
def test_conv2d_vs_fp64_and_repeatable(geo):
    from pleas_merging_amd import hip_ops

    N, Cin, H, W, Cout, k, stride, pad, has_bias = geo
    g = torch.Generator().manual_seed(sum(geo))
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() if has_bias else None
    want = F.conv2d(x.double(), w.double(), b.double() if has_bias else None, stride, pad)
    ref = F.conv2d(x, w, b, stride, pad)
    kpos = k > 1 and Cin % 32 == 0
    wk = w.permute(0, 2, 3, 1).contiguous() if kpos else w
    got = hip_ops.conv2d(x, wk, b, stride, pad, kpos)
    assert got.shape == want.shape
    rel = lambda a: float((a.double() - want).norm() / want.norm())
    print("conv2d %s: own %.2e, vendor %.2e from fp64" % (geo, rel(got), rel(ref)))
    assert rel(got) <= max(2e-6, 3 * rel(ref)), (rel(got), rel(ref))
    for _ in range(4):      # fixed summation order: the same bits every time
        assert torch.equal(hip_ops.conv2d(x, wk, b, stride, pad, kpos), got)


@pytest.mark.parametrize("arith", ["fp32", "split_bf16"])
@pytest.mark.parametrize("geo", GEOMETRIES + [(2, 64, 56, 56, 256, 1, 1, 0, False),      # short-K 1 x 1, 128-row tiles
                                              (3, 2048, 7, 7, 512, 1, 1, 0, False)])     # 1 x 1 on 7 x 7 images
def test_conv2d_bn_act_equals_the_two_launches_it_replaces(geo, arith):
    """``pleas_conv2d_bn_act_fwd``: the convolution's output AND its BatchNorm / add / ReLU image from one launch -- both
    ``torch.equal`` to ``conv2d`` followed by ``bn_act`` (with and without identity, with and without ReLU), every tile form,
    both arithmetics."""
    from pleas_merging_amd import hip_ops

    N, Cin, H, W, Cout, k, stride, pad, has_bias = geo
    g = torch.Generator().manual_seed(sum(geo) + 1)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() if has_bias else None
    scale = (0.5 + torch.rand(Cout, generator=g)).cuda()
    shift = torch.randn(Cout, generator=g).cuda()
    kpos = k > 1 and Cin % 32 == 0
    wk = w.permute(0, 2, 3, 1).contiguous() if kpos else w
    from pleas_merging_amd import _lib

    lib = _lib.lib()
    assert lib.pleas_arith_get() == 0
    lib.pleas_arith(1 if arith == "split_bf16" else 0)
    try:
        y0 = hip_ops.conv2d(x, wk, b, stride, pad, kpos)
        res = torch.randn(y0.shape, generator=g).cuda()
        for identity, relu in ((None, True), (res, True), (None, False), (res, False)):
            y, z = hip_ops.conv2d_bn_act(x, wk, b, stride, pad, kpos, scale, shift, identity, relu)
            assert torch.equal(y, y0), (geo, "y")
            assert torch.equal(z, hip_ops.bn_act(y0, scale, shift, identity, relu)), (geo, identity is not None, relu)
    finally:
        lib.pleas_arith(0)


def test_source_forward_is_repeatable_and_matches_the_modules():
    """One frozen-source forward of a ResNet-101 twice: every tap bit-equal; and against the model's own modules (vendor
    convolutions + vendor BatchNorm) to fp32 rounding."""
    from pleas_merging_amd import resnet as zoo
    from pleas_merging_amd.methods.pleas_merging import FrozenSources
    from pleas_merging_amd.methods.source_forward import HipConv, HipConvBnAct

    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 3, 224, 224, generator=g).cuda()
    models = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = zoo.MODELS["resnet101"](num_classes=1000).cuda()
        zoo.calibrate_bn(m, [x])
        models.append(m.eval())
    src = FrozenSources(*models)
    calls = [n.target for n in src.src1.graph.nodes if n.op == "call_function"]
    own = sum(1 for t in calls if isinstance(t, (HipConv, HipConvBnAct)))
    assert own == 104, own         # every convolution of a ResNet-101 (SOURCE_CONV = "all"); the classifier stays a vendor GEMM
    # ... and all but the stem (whose chain ends in the pooling pass) carry their BatchNorm chain in the epilogue
    assert sum(1 for t in calls if isinstance(t, HipConvBnAct)) == 103
    runs = []
    for _ in range(3):
        _, (in1, out1), (in2, out2), _ev = src.launch(x)
        torch.cuda.synchronize()
        runs.append(({k: v.clone() for k, v in out1.items()}, {k: v.clone() for k, v in out2.items()}))
    for a, b in zip(runs[0], runs[1]):
        assert len(a) == 105
        for k in a:
            assert torch.equal(a[k], b[k]), k
    src.close()
    # accuracy: 101 layers amplify rounding differences (layer 4 of two fp32 forwards of this model differs by 1e-4..1e-3), so
    # both fp32 paths are held to an fp64 forward of the same model: the fused path (own k x k convolutions, folded BatchNorm)
    # must be as close to it, layer by layer, as the model's own modules (vendor convolutions, vendor BatchNorm) are
    import copy

    def taps_of(model, inp):
        got = {}
        hooks = [mod.register_forward_hook(lambda m_, i, o, n=n: got.__setitem__(n, o))
                 for n, mod in model.named_modules() if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear))]
        with torch.no_grad():
            model(inp)
        for h in hooks:
            h.remove()
        return got

    vendor = taps_of(models[0], x)
    exact = taps_of(copy.deepcopy(models[0]).double(), x.double())
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    worst = (0.0, 0.0, "")
    for k in exact:
        own_err, ven_err = rel(runs[0][0][k], exact[k]), rel(vendor[k], exact[k])
        assert own_err <= max(3 * ven_err, 2e-6), (k, own_err, ven_err)
        worst = max(worst, (own_err, ven_err, k))
    print("fused source forward vs fp64, worst layer %s: %.2e (the model's own modules there: %.2e)" % (worst[2], worst[0], worst[1]))


def test_fused_epilogue_taps_equal_the_two_launch_graph(monkeypatch):
    """The frozen-source graph with the BatchNorm chains in the convolutions' epilogues (default) against the graph that runs
    ``conv2d`` and ``bn_act`` as two launches (``SOURCE_CONV_BN = "0"``): every hooked input and output ``torch.equal``, on a
    ResNet-50 at two batch sizes (the second one with pixel tiles that straddle samples)."""
    from pleas_merging_amd import resnet as zoo
    from pleas_merging_amd.methods import source_forward
    from pleas_merging_amd.methods.pleas_merging import FrozenSources
    from pleas_merging_amd.methods.source_forward import HipConvBnAct

    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(n, 3, 224, 224, generator=g).cuda() for n in (4, 3)]
    models = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = zoo.MODELS["resnet50"](num_classes=100).cuda()
        zoo.calibrate_bn(m, [xs[0]])
        models.append(m.eval())

    def taps(fused):
        monkeypatch.setattr(source_forward, "SOURCE_CONV_BN", "1" if fused else "0")
        src = FrozenSources(*models)
        n = sum(1 for nd in src.src1.graph.nodes if nd.op == "call_function" and isinstance(nd.target, HipConvBnAct))
        assert n == (52 if fused else 0), n
        out = []
        for x in xs:
            _, (in1, out1), (in2, out2), _ev = src.launch(x)
            torch.cuda.synchronize()
            out.append([{k: v.clone() for k, v in t.items()} for t in (in1, out1, in2, out2)])
        src.close()
        return out

    a, b = taps(True), taps(False)
    for per_x_a, per_x_b in zip(a, b):
        for ta, tb in zip(per_x_a, per_x_b):
            assert ta.keys() == tb.keys() and len(ta) == 54
            for k in ta:
                assert torch.equal(ta[k], tb[k]), k


def test_rn101_job_twice_is_bit_identical(tmp_path):
    """The ResNet-101 job of test_hip_fullsize_dp (matching over 2 batches -> 71 LAPs -> partial merge at ratio 0.5 -> 5 updates
    with grouped source forwards) run twice in one process: costs, assignments, trained weights and losses ``torch.equal``."""
    path = os.path.join(str(tmp_path), "inputs.pt")
    dp._make_inputs(path)
    a = dp._job(path, data_parallel=False)
    b = dp._job(path, data_parallel=False)
    for k in a["costs"]:
        assert torch.equal(a["costs"][k], b["costs"][k]), k
        assert torch.equal(a["perm"][k], b["perm"][k]), k
    for k, v in a["sd"].items():
        assert torch.equal(v, b["sd"][k]), k
    assert torch.equal(a["loss"], b["loss"])
