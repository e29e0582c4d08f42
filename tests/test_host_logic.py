"""CPU tests (no GPU): host-side logic of the drop-in API, the C-ABI surface, loud failure."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO
from oracle import pleas_oracle as orc
from pleas_merging_amd.core.utils import Axis, PermutationGroup, spec_to_json


# ------------------------------------------------------------------ spec builder (G1)
@pytest.mark.parametrize("name", ["resnet18", "resnet50", "resnet101", "resnet50_identity_fc"])
def test_spec_matches_reference_fixture(name):
    from pleas.core.compiler import get_permutation_spec
    from pleas_merging_amd import resnet as zoo

    m = getattr(zoo, name.split("_")[0])()
    if name.endswith("identity_fc"):
        m.fc = torch.nn.Identity()
    spec = get_permutation_spec(m, ((1, 3, 224, 224),))
    gold = json.load(open(os.path.join(GOLDEN, "spec_%s.json" % name)))["spec"]
    assert spec_to_json(spec) == gold  # same keys, same ORDER, same sizes / state / node sets


def test_spec_function_invariance(tiny_bottleneck):
    from pleas.core.compiler import check_permutation_spec, get_permutation_spec

    m = tiny_bottleneck.m1
    spec = get_permutation_spec(m, ((2, 3, 32, 32),))
    assert spec_to_json(spec) == spec_to_json(tiny_bottleneck.spec)
    assert check_permutation_spec(m, spec, torch.randn(2, 3, 32, 32))


def test_unsupported_op_raises():
    from pleas.core.compiler import get_permutation_spec

    class Odd(torch.nn.Module):
        def forward(self, x):
            return torch.cumsum(x, 1)

    with pytest.raises(NotImplementedError):
        get_permutation_spec(Odd(), ((2, 4),))


# ------------------------------------------------------------------ types / permutations
def test_axis_and_perm_helpers(tiny_basic):
    from pleas.core.utils import apply_perm, invert_perm, make_identity_perm, make_random_perm, perm_eq

    assert str(Axis("layer1.0.conv1.weight", 0)) == "layer1.0.conv1.weight:0"
    assert Axis("a", 1) == Axis("a", 1) and ("a", 1) not in {Axis("a", 1): 0}  # the reference's F2 trap
    spec = tiny_basic.spec
    p = make_random_perm(spec, torch.Generator().manual_seed(0))
    assert perm_eq(invert_perm(invert_perm(p)), p)
    sd = tiny_basic.m1.state_dict()
    back = apply_perm(invert_perm(p), spec, apply_perm(p, spec, sd))
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    assert perm_eq(make_identity_perm(spec), {k: torch.arange(g.size) for k, g in spec.items()})
    with pytest.raises(AssertionError):
        apply_perm(p, spec, tiny_basic.m1, inplace=False)


# ------------------------------------------------------------------ plug points on CPU callables
def test_activation_matching_generic_path_cpu(tiny_bottleneck):
    from pleas.core.solvers import scipy_solve_lsa
    from pleas.methods.activation_matching import activation_matching

    t = tiny_bottleneck
    perm, costs = activation_matching(t.spec, t.m1, t.m2, t.batches(), 3, cross_features=orc.cross_features_cdist,
                                      lsa_solver=scipy_solve_lsa, output_costs=True, accumulate="reference")
    for k in t.spec:
        assert (perm[k] == t.per_key("am_perm")[k]).all()
        assert torch.allclose(costs[k], t.per_key("am_cost")[k], rtol=1e-5, atol=1e-5)


def test_weight_matching_host_loop_cpu(tiny_basic):
    from pleas.core.solvers import scipy_solve_lsa
    from pleas.methods.weight_matching import weight_matching

    t = tiny_basic
    perm, costs = weight_matching(t.spec, t.m1.state_dict(), t.m2.state_dict(), max_iter=100, seed=0, verbose=False,
                                  lsa_solver=scipy_solve_lsa, cross_weights=orc.cross_features_inner_product,
                                  return_costs=True)
    for k in t.spec:
        assert (perm[k] == torch.from_numpy(t.z["wm_perm/%s" % k])).all()
        assert torch.allclose(costs[k], torch.from_numpy(t.z["wm_cost/%s" % k]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("ratio", [0.0, 0.5, 1.0])
def test_get_blocks_cpu_tensors(tiny_basic, ratio):
    from pleas.methods.partial_matching import block_maps, get_blocks

    t = tiny_basic
    blocks = get_blocks(t.spec, t.per_key("am_perm"), t.per_key("am_cost"), ratio, False)
    tag = "r%03d" % int(ratio * 100)
    for k in t.spec:
        for j in range(4):
            assert (blocks[k][j] == torch.from_numpy(t.z["blocks_%s/%s/%d" % (tag, k, j)])).all()
        r1, r2, nm = block_maps(blocks[k], "cpu")
        assert nm == len(blocks[k][0]) and len(r1) == len(r2) == nm + 2 * len(blocks[k][2])
        assert (r1[nm:nm + len(blocks[k][2])] >= 0).all() and (r1[nm + len(blocks[k][2]):] == -1).all()


def test_gradient_mask_matches_oracle(tiny_basic):
    from pleas.methods.partial_matching import get_blocks, spread_blocks
    from pleas.methods.pleas_merging import get_gradient_mask

    t = tiny_basic
    blocks = spread_blocks(t.spec, get_blocks(t.spec, t.per_key("am_perm"), t.per_key("am_cost"), 0.5, False))
    m3 = orc.partial_merge(t.spec, t.m1, t.m2, t.per_key("am_perm"), t.per_key("am_cost"), 0.5)
    layers = {n: m for n, m in m3.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))}
    got, want = get_gradient_mask(blocks, layers), orc.gradient_masks(blocks, layers)
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    assert any((m == 0).any() for m in got)  # ratio 0.5 really freezes blocks


def test_cosine_lrs_equal_torch_scheduler():
    from pleas.methods.pleas_merging import cosine_lrs  # noqa: F401
    from pleas_merging_amd.methods.pleas_merging import cosine_lrs

    for t_max in (1, 5, 20, 400):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=5e-4)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, t_max)
        want = []
        for _ in range(t_max + 1):
            want.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        assert cosine_lrs(5e-4, t_max, t_max + 1) == want


# ------------------------------------------------------------------ C-ABI surface
def _declared_symbols():
    text = open(os.path.join(REPO, "include", "pleas_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pleas_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pleas_merging_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(handle, name), "libpleas_hip.so lacks %s declared in include/pleas_hip.h" % name
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)
    handle.pleas_version.restype = ctypes.c_char_p
    assert b"gfx950" in handle.pleas_version()


def test_normal_eq_plan_contracts_one_block_per_lag_class():
    """Host side of pleas_normal_eq_accum (no GPU): a stride-1 3x3 layer contracts 29 of the 45 blocks of its lower
    triangle -- one per lag class up to transposition -- and leaves 16 to pleas_normal_eq_finalize; 1x1 and strided
    layers contract every block.  ResNet-101's layer3 conv2 (K = 2304) executes 0.66 of the triangle's flops."""
    from pleas_merging_amd import _lib

    lib = _lib.lib()
    info = (ctypes.c_double * 4)()

    def plan(N, C, H, W, k, stride, pad):
        arr = (_lib.NeqLayer * 1)()
        arr[0].N, arr[0].Cin, arr[0].Hin, arr[0].Win, arr[0].KH, arr[0].KW, arr[0].stride, arr[0].pad = N, C, H, W, k, k, stride, pad
        assert lib.pleas_normal_eq_plan_info(arr, 1, info) == 0
        return list(info)

    flops, executed, items, copies = plan(16, 256, 14, 14, 3, 1, 1)
    assert flops == 2304.0 ** 2 * 16 * 196 and copies == 16 and items == 9 * 3 + 20 * 4
    assert abs(executed / flops - 107 * 2 * 128 * 128 * 98 * 32 / flops) < 1e-9 and 0.65 < executed / flops < 0.67
    assert plan(16, 1024, 14, 14, 1, 1, 0)[3] == 0 and plan(16, 128, 56, 56, 3, 2, 1)[3] == 0
    assert plan(4, 24, 8, 8, 5, 1, 2)[3] == 168
    os.environ["PLEAS_NEQ_LAG"] = "0"
    try:
        assert plan(16, 256, 14, 14, 3, 1, 1)[3] == 0
    finally:
        del os.environ["PLEAS_NEQ_LAG"]
    assert lib.pleas_normal_eq_plan_info(None, 0, info) == -22 and lib.pleas_normal_eq_finalize(None, 0, None) == -22


def test_weight_matching_independent_runs_equal_the_sequential_sweep(tiny_basic, tiny_bottleneck):
    """On the device consecutive visits that do not depend on each other share one grouped contraction and one LAP launch
    (``_Visitor.independent_run``): no group of a run scores a tensor that an earlier group of the run permutes, so every
    score matrix is what the reference's one-by-one sweep (weight_matching.py:59-91) computes at that visit.  Checked on
    the host through the plug-point path: (1) the rule itself on the ResNet-101 spec, (2) a sweep in runs gives the same
    permutations and score matrices as the sweep one visit at a time, bit for bit."""
    import importlib

    from pleas_merging_amd.core.utils import spec_from_json

    wm = importlib.import_module("pleas_merging_amd.methods.weight_matching")    # the package re-exports the FUNCTION by that name

    host_inner = lambda a, b, axis: torch.einsum("ik,jk->ij", a.movedim(axis, 0).reshape(a.shape[axis], -1),
                                                 b.movedim(axis, 0).reshape(b.shape[axis], -1))
    # (1) the independence rule on a big spec, random orders
    spec = spec_from_json(json.load(open(os.path.join(GOLDEN, "spec_resnet101.json")))["spec"])
    keys = sorted({ax.key for g in spec.values() for ax in g.state})
    sd = {k: torch.zeros(1) for k in keys}
    walker = wm._Visitor(spec, [sd], [dict(sd)], ("running_mean", "running_var"), True, host_inner, orc.solve_lsa, batch_runs=True)
    names = list(spec.keys())
    rng = torch.Generator().manual_seed(0)
    lengths = []
    for _ in range(5):
        order = [names[i] for i in torch.randperm(len(names), generator=rng)]
        at = 0
        while at < len(order):
            end = walker.independent_run(order, at)
            assert end > at
            for j in range(at, end):
                scored = {ax.key for _, ax in walker.axes[order[j]]}
                for i in range(at, j):
                    assert not scored & {ax.key for ax in spec[order[i]].state}, (order[i], order[j])
            lengths.append(end - at)
            at = end
    assert sum(lengths) == 5 * len(names) and max(lengths) >= 4      # most visits of a ResNet sweep do share launches
    # (2) same result as one visit at a time
    for t in (tiny_basic, tiny_bottleneck):
        outs = []
        for runs in (False, True):
            real = wm._Visitor
            wm._Visitor = lambda *a, _real=real, _runs=runs, **k: _real(*a, batch_runs=_runs, **k)
            try:
                outs.append(wm.weight_matching(t.spec, t.m1.state_dict(), t.m2.state_dict(), max_iter=20, seed=0, verbose=False,
                                               lsa_solver=orc.solve_lsa, cross_weights=host_inner, return_costs=True))
            finally:
                wm._Visitor = real
        (p0, c0), (p1, c1) = outs
        for k in t.spec:
            assert torch.equal(p0[k], p1[k]) and torch.equal(c0[k], c1[k]), k


def test_forward_launch_units_partition_every_form():
    """Host side of pleas_fwd_batch (no GPU): the grouped forward is launched as UNITS -- a tile form over a slice of its
    items, on one of four lanes.  A fresh plan has one unit per form; after a calibration launch a form that outlasts a
    lane's fair share is cut into slices of equal work.  Whatever the measured durations, the units must partition every
    form's item range exactly (a missing or doubled slice would drop or repeat output tiles) and use lanes 0..3."""
    from pleas_merging_amd import _lib

    sys.path.insert(0, os.path.join(REPO, "tests"))
    from sanitize_driver_layers import resnet_layers

    lib = _lib.lib()
    layers = resnet_layers("resnet101")
    arr = (_lib.FwdLayer * len(layers))()
    for f, (co, ci, h, w, k, s_, p_) in zip(arr, layers):
        f.N, f.Cout, f.Cin, f.Hin, f.Win, f.KH, f.KW, f.stride, f.pad = 16, co, ci, h, w, k, k, s_, p_
        f.Csrc, f.n_merged, f.flags = co, co, (1 if (k > 1 and ci % 32 == 0) else 0)
    units = (ctypes.c_int * (4 * 24))()

    def get(ms):
        n = lib.pleas_fwd_plan_units(arr, len(layers), ms, units, 24)
        assert 0 < n <= 24, n
        return [tuple(units[4 * i:4 * i + 4]) for i in range(n)]

    fresh = get(None)
    forms = {u[0]: (u[1], u[2]) for u in fresh}
    assert len(forms) == len(fresh) and sum(c for _, c in forms.values()) > 15000      # one unit per form (round 5: 128-row tiles
    # for the short-K 1 x 1 layers too: 18 036 items, was 27 732)
    rng = np.random.default_rng(0)
    cases = [np.array([1.76, 0, 0, 0.24, 2.56, 0.27, 1.61, 2.56, 0, 0.36]),             # measured in round 3's bench job
             np.ones(10), np.array([0, 0, 0, 0, 0, 0, 0, 9.0, 0, 0.01])] + [rng.random(10) * 3 for _ in range(20)]
    for ms in cases:
        got = get((ctypes.c_double * 10)(*ms.tolist()))
        assert all(0 <= lane <= 3 for _, _, _, lane in got) and got[0][3] == 0
        for f, (begin, count) in forms.items():
            slices = sorted((b, c) for ff, b, c, _ in got if ff == f)
            assert slices and slices[0][0] == begin and sum(c for _, c in slices) == count, (f, slices, begin, count)
            for (b0, c0), (b1, _) in zip(slices, slices[1:]):
                assert b0 + c0 == b1 and c0 > 0, (f, slices)
    split = get((ctypes.c_double * 10)(*cases[0].tolist()))
    assert len([u for u in split if u[0] == 4]) >= 2   # the long 1 x 1 form is cut into slices


def test_host_code_under_sanitizers():
    """SURVEY.md section 5 (sanitizers): the host halves of csrc/*.hip -- plan builders, XCD item ordering, lane dealing,
    lag classes, the host LAP, argument checks -- compiled with -fsanitize=address,undefined (device code as always) and
    driven without a GPU through the `*_ws_bytes` / `*_plan_info` / `pleas_lsap_host` entry points on the ResNet-18 / 50 /
    101 layer lists and on degenerate sizes (tests/sanitize_driver.py, a torch-free subprocess with the ASan runtime
    preloaded).  Any report aborts the driver."""
    import subprocess

    from pleas_merging_amd import build

    lib = build.build_sanitized()
    syms = subprocess.check_output(["nm", "-D", lib], text=True)
    assert "__asan_report" in syms and "__ubsan_handle" in syms          # the host code really is instrumented
    env = dict(os.environ, LD_PRELOAD=build.asan_runtime(), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "sanitize_driver.py"), lib], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SANITIZE_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


def test_argument_errors_without_gpu():
    """Host-side validation returns error codes before anything touches a device."""
    from pleas_merging_amd import _lib

    lib = _lib.lib()
    assert lib.pleas_gram_accum(None, None, 1, 4, 4, 0, 0, None, None, 0, None) == -22
    assert b"null" in lib.pleas_last_error()
    assert lib.pleas_gram_ws_bytes(16, 256, 196) >= 256 * 256 * 4
    assert lib.pleas_gram_ws_bytes(0, 256, 196) == 0
    n = (ctypes.c_int * 1)(4097)
    ptr = (ctypes.c_void_p * 1)(8)
    assert lib.pleas_lsap_batched(ptr, n, 1, 1, ptr, None) == -22  # n > PLEAS_LSAP_MAX_N
    assert lib.pleas_masked_adam(None, None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 1, None) == -22


def test_product_path_fails_loudly_on_cpu(tiny_basic):
    from pleas.methods.activation_matching import activation_matching, cross_features_cdist
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import train
    from pleas.methods.weight_matching import weight_matching
    from pleas_merging_amd.hip_ops import PleasHipError

    t = tiny_basic
    with pytest.raises(PleasHipError):
        cross_features_cdist(torch.randn(2, 4, 3, 3), torch.randn(2, 4, 3, 3), 1)
    with pytest.raises(RuntimeError):
        activation_matching(t.spec, t.m1, t.m2, t.batches(), 1)
    with pytest.raises(PleasHipError):
        weight_matching(t.spec, t.m1.state_dict(), t.m2.state_dict(), verbose=False)
    if not torch.cuda.is_available():
        with pytest.raises((PleasHipError, RuntimeError, AssertionError)):
            partial_merge(t.spec, t.m1, t.m2, t.per_key("am_perm"), t.per_key("am_cost"), 0.5)
    m3 = orc.partial_merge(t.spec, t.m1, t.m2, t.per_key("am_perm"), t.per_key("am_cost"), 0.5)
    with pytest.raises(PleasHipError):
        train(t.batches("xt"), t.m1, t.m2, m3, t.spec, t.per_key("am_perm"), t.per_key("am_cost"), 0.5, False, 2, None,
              num_classes=10)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pleas_merging_amd/ or pleas/ may import it."""
    for root in ("pleas_merging_amd", "pleas"):
        for dirpath, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "oracle" not in text.replace("oracle/", "").lower() or f == "build.py", os.path.join(dirpath, f)


# ------------------------------------------------------------------ section 8(f) "next" rows (host logic)
def test_zip_ratios_stage_rule():
    from pleas.core.compiler import get_permutation_spec
    from pleas.methods.extras import zip_ratios
    from pleas_merging_amd import resnet as zoo

    spec = get_permutation_spec(zoo.resnet50(), ((1, 3, 224, 224),))
    for budget, last in ((1.0, 4), (1.2, 3), (1.55, 2), (1.8, 1), (2.0, 0)):
        r = zip_ratios(spec, budget)
        assert set(r) == set(spec)
        for k, v in r.items():
            stage = int(k.key.split(".")[0][5:]) if k.key.startswith("layer") else 0
            assert v == (0.0 if stage <= last else 1.0), (budget, k)
    with pytest.raises(KeyError):
        zip_ratios(spec, 1.3)


def test_matching_and_checkpoint_round_trip(tiny_basic, tmp_path):
    from pleas.methods.extras import load_checkpoint, load_matching, save_matching
    from pleas_merging_amd import resnet as zoo

    t = tiny_basic
    perm, costs = t.per_key("am_perm"), t.per_key("am_cost")
    save_matching(str(tmp_path / "m.pt"), perm, costs)
    p2, c2 = load_matching(str(tmp_path / "m.pt"))
    assert list(p2) == list(perm) and all(torch.equal(p2[k], perm[k]) and torch.equal(c2[k], costs[k]) for k in perm)
    for blob in (t.m1.state_dict(), {"model": t.m1.state_dict(), "epoch": 3}):
        torch.save(blob, str(tmp_path / "ck.pt"))
        m = load_checkpoint(zoo.tiny_resnet("basic", (1, 1, 1, 1), 10, 4), str(tmp_path / "ck.pt"))
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), t.m1.state_dict().values()))


def test_reset_bn_stats_matches_driver_procedure(tiny_basic):
    import copy

    from pleas.methods.extras import reset_bn_stats

    t = tiny_basic
    m3 = orc.partial_merge(t.spec, t.m1, t.m2, t.per_key("am_perm"), t.per_key("am_cost"), 0.5)
    ref = copy.deepcopy(m3).train()   # the drivers' procedure, written out (run_domainnet.py:327-341)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.reset_running_stats()
    data = t.batches("xt")
    with torch.no_grad():
        for i, b in enumerate(data):
            ref(b[0].float())
            if i + 1 > 100:
                break
    got = reset_bn_stats(m3, data, 101)
    assert got.training
    for (k, a), (_, b) in zip(got.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k


def test_source_forward_rewrite_keeps_values_and_hooks(tiny_bottleneck):
    """fx rewrite of the frozen sources (BN -> [+identity] -> ReLU => one op): checked on CPU with a torch stand-in
    for the HIP op -- same logits, same hooked conv inputs/outputs, no BatchNorm/ReLU module call left."""
    from pleas_merging_amd.methods.source_forward import fuse_bn_act

    def op(x, s, t, res, relu):
        y = x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
        y = y if res is None else y + res
        return torch.relu(y) if relu else y

    import copy

    model = copy.deepcopy(tiny_bottleneck.m1).eval()
    gm = fuse_bn_act(model, op)
    assert gm is not None
    mods = dict(gm.named_modules())
    calls = [n for n in gm.graph.nodes if n.op == "call_function" and n.target is op]
    assert any(n.args[3] is not None for n in calls) and any(not n.args[4] for n in calls)
    assert not [n for n in gm.graph.nodes
                if n.op == "call_module" and isinstance(mods[n.target], (torch.nn.BatchNorm2d, torch.nn.ReLU))]
    seen = {}
    handles = [m.register_forward_hook(lambda m, i, o, n=n: seen.setdefault(n, []).append((i[0].clone(), o.clone())))
               for n, m in model.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
    x = tiny_bottleneck.batches()[0][0]
    with torch.no_grad():
        a, b = model(x), gm(x)
    for h in handles:
        h.remove()
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
    for name, (first, second) in seen.items():
        assert torch.allclose(first[0], second[0], rtol=1e-4, atol=1e-5), name
        assert torch.allclose(first[1], second[1], rtol=1e-4, atol=1e-5), name
    # a BN -> ReLU chain consumed by a max pooling alone (a ResNet's stem) becomes ONE pooled op -- only when the caller
    # supplies it next to its own `op`
    from pleas_merging_amd import resnet as zoo

    def pool(x, s, t, kernel, stride, padding, relu):
        return torch.nn.functional.max_pool2d(op(x, s, t, None, relu), kernel, stride, padding)

    torch.manual_seed(3)
    rn = zoo.resnet18(num_classes=10).eval()
    for m in rn.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
    plain, pooled = fuse_bn_act(rn, op), fuse_bn_act(rn, op, pool_op=pool)
    assert [n.target for n in plain.graph.nodes if n.op == "call_module" and n.target == "maxpool"] == ["maxpool"]
    calls = [n for n in pooled.graph.nodes if n.op == "call_function" and n.target is pool]
    assert len(calls) == 1 and calls[0].args[3:] == ((3, 3), 2, 1, True)
    assert not [n for n in pooled.graph.nodes if n.op == "call_module" and n.target == "maxpool"]
    xs = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        assert torch.allclose(rn(xs), pooled(xs), rtol=1e-4, atol=1e-5)
    rn.maxpool.ceil_mode = True                 # a window the kernel does not compute: left to the module
    assert not [n for n in fuse_bn_act(rn, op, pool_op=pool).graph.nodes if n.op == "call_function" and n.target is pool]
    model.train()
    assert fuse_bn_act(model, op) is None   # training-mode BN updates running stats: never folded
    model.eval()


def test_source_graph_joins_convolutions_with_their_batchnorm_chain(monkeypatch):
    """Graph structure of the default frozen-source rewrite (no GPU): every dense square Conv2d becomes an own-kernel call, and a
    convolution whose ONLY consumer is an eval-mode BatchNorm chain carries that chain (``HipConvBnAct``: conv -> bn -> relu,
    conv3 -> bn3 -> + identity -> relu, downsample conv -> bn); the stem, whose chain ends in the pooling pass, a convolution
    with a second consumer, and ``SOURCE_CONV_BN = "0"`` keep two calls."""
    from pleas_merging_amd import resnet as zoo
    from pleas_merging_amd.methods import source_forward as sf

    torch.manual_seed(5)
    rn = zoo.resnet50(num_classes=10).eval()
    calls = lambda gm: [n.target for n in gm.graph.nodes if n.op == "call_function"]
    gm = sf.fuse_bn_act(rn)
    fused = [t for t in calls(gm) if isinstance(t, sf.HipConvBnAct)]
    assert len(fused) == 52 and sum(1 for t in calls(gm) if isinstance(t, sf.HipConv)) == 1          # 53 convolutions, the stem alone
    assert not [n for n in gm.graph.nodes if n.op == "call_module" and isinstance(rn.get_submodule(n.target), torch.nn.Conv2d)]
    assert sum(1 for t in calls(gm) if t is sf._bn_act) == 0
    # identity operands: 16 bottleneck blocks end in conv3 -> bn3 -> (+ identity) -> relu
    with_res = [n for n in gm.graph.nodes if n.op == "call_function" and isinstance(n.target, sf.HipConvBnAct) and n.args[3] is not None]
    assert len(with_res) == 16 and all(n.args[4] is True for n in with_res)
    # the same module objects: hooks registered on the model fire from the rewritten graph
    assert all(t.own.conv is rn.get_submodule(name) for t, name in [(fused[0], "layer1.0.conv1")])
    monkeypatch.setattr(sf, "SOURCE_CONV_BN", "0")
    two = sf.fuse_bn_act(rn)
    assert not [t for t in calls(two) if isinstance(t, sf.HipConvBnAct)]
    assert sum(1 for t in calls(two) if isinstance(t, sf.HipConv)) == 53 and sum(1 for t in calls(two) if t is sf._bn_act) == 52


def test_split_twin_graph_defers_sinks_and_keeps_inplace_targets(tiny_bottleneck):
    """The two-stream twin graph (model2's chain, then model1's, then the sinks) sees the same activations as the
    interleaved one -- also for a model that overwrites tracked tensors in place (``out += identity``, ``relu_``)."""
    import importlib

    am = importlib.import_module("pleas_merging_amd.methods.activation_matching")

    class Streams:   # stand-in for _SideStream on a machine without a GPU
        def __init__(self):
            self.log = []

            def fork(x):
                self.log.append("fork")
                return x

            def back():
                self.log.append("back")

            def join():
                self.log.append("join")

            for f in (fork, back, join):
                f.__qualname__ = f.__name__
            self.fork, self.back, self.join = fork, back, join

    def emitter(store):
        def emit(g, name, a, n1, n2):
            def sink(x, y, a, _n=name):
                store[_n, a] = (x.clone(), y.clone())

            sink.__name__ = sink.__qualname__ = "sink_" + name
            return g.call_function(sink, (n1, n2, a))

        return emit

    class InPlace(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.c2 = torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Conv2d(4, 4, 3, padding=1)
            self.act = torch.nn.ReLU(inplace=True)

        def forward(self, x):
            a = self.act(self.c1(x))
            b = self.c2(a)
            b += a
            return b.relu_()

    t = tiny_bottleneck
    cases = [(t.m1, t.m2, [nax for g in t.spec.values() for nax in g.node], t.batches()[0][0])]
    torch.manual_seed(0)
    cases.append((InPlace(), InPlace(), [Axis("c1", 1), Axis("c2", 1), Axis("act", 1)], torch.randn(2, 3, 6, 6)))
    for m1, m2, axes, x in cases:
        inter, split, streams = {}, {}, Streams()
        with torch.no_grad():
            am._build_twin(m1, m2, axes, emitter(inter), keep_inputs=True)(x)
            out = am._build_twin(m1, m2, axes, emitter(split), keep_inputs=True, side_stream=streams)(x)
            want = (m1(x.clone()), m2(x.clone()))
        assert streams.log == ["fork", "back", "join"]
        assert torch.equal(out[0][0], want[0]) and torch.equal(out[0][1], want[1])
        assert set(inter) == set(split) and len(split) == len(set(axes))
        for k in inter:
            assert torch.equal(inter[k][0], split[k][0]) and torch.equal(inter[k][1], split[k][1]), k
    # the in-place model: what the sinks saw is what a hook would have seen right after each node
    seen = {}
    m1, m2, axes, x = cases[1]
    hooks = [m1.c1.register_forward_hook(lambda m, i, o: seen.__setitem__("c1", o.clone())),
             m1.c2.register_forward_hook(lambda m, i, o: seen.__setitem__("c2", o.clone()))]
    with torch.no_grad():
        m1(x)
    for h in hooks:
        h.remove()
    assert torch.equal(split["c1", 1][0], seen["c1"]) and torch.equal(split["c2", 1][0], seen["c2"])
    with pytest.raises(ValueError):
        am._build_twin(m1, m2, axes, emitter({}), keep_inputs=False, side_stream=Streams())



@pytest.mark.parametrize("fname", ["tiny_basic.npz", "tiny_bottleneck.npz"])
def test_flop_model_and_fc_helpers_match_reference_fixture(fname):
    """SURVEY 8(f) rows 2 and 4 against vectors produced by the reference (tests/golden/make_golden_extras.py)."""
    from conftest import Tiny
    from pleas.core.utils import count_linear_flops
    from pleas.methods.partial_matching import partial_merge_flops
    from pleas.methods.pleas_merging import get_fc_perm, permute_final_features

    t = Tiny(fname)
    case = json.load(open(os.path.join(GOLDEN, "extras_tiny.json")))["cases"][fname]
    flops, terms = count_linear_flops(t.spec, t.m1, ((2, 3, 32, 32),))
    assert flops == case["flops"]
    assert [[float(c)] + [str(a) for a in axes] for c, *axes in terms] == case["terms"]
    for entry in case["merge_flops"]:
        r = entry["ratios"]
        r = {Axis.parse(k): v for k, v in r.items()} if isinstance(r, dict) else r
        assert abs(partial_merge_flops(t.spec, terms, r) - entry["flops"]) <= 1e-9 * entry["flops"]
    perm, costs = t.per_key("am_perm"), t.per_key("am_cost")
    for entry in case["fc"]:
        blocks = get_fc_perm(perm, t.spec, costs, entry["ratio"])
        assert [b.tolist() for b in blocks] == entry["blocks"]
        feats = torch.tensor(entry["features"])
        for idx in (0, 1):
            assert torch.equal(permute_final_features(feats, blocks, idx), torch.tensor(entry["permuted"][idx]))


def test_qp_ratios_without_gurobi(tiny_basic):
    """The reference returns random ratios without Gurobi; ours must be feasible, use the budget, be monotone in it
    and hit the optimum of a small instance found by exhaustive search."""
    from pleas.core.utils import count_linear_flops
    from pleas.methods.partial_matching import partial_merge_flops, qp_ratios

    t = tiny_basic
    _, terms = count_linear_flops(t.spec, t.m1, ((2, 3, 32, 32),))
    base = partial_merge_flops(t.spec, terms, 0.0)
    full = partial_merge_flops(t.spec, terms, 1.0) / base
    w = {k: 0.5 + 0.25 * i for i, k in enumerate(t.spec)}
    assert all(v == 0.0 for v in qp_ratios(t.spec, terms, 1.0, w).values())
    assert all(v == 1.0 for v in qp_ratios(t.spec, terms, full + 0.1, w).values())
    last = -1.0
    for budget in (1.1, 1.3, 1.6, 1.9):
        r = qp_ratios(t.spec, terms, budget, w)
        assert set(r) == set(t.spec) and all(0.0 <= v <= 1.0 for v in r.values())
        used = partial_merge_flops(t.spec, terms, r) / base
        assert used <= budget + 1e-9 and used >= min(budget, full) - 1e-3      # feasible and budget-tight
        obj = sum(w[k] * r[k] for k in r)
        assert obj >= last - 1e-9
        last = obj
        assert qp_ratios(t.spec, terms, budget, w) == r                          # deterministic
    # exhaustive check on a two-group toy: F = 10 (1 + a) + 6 (1 + a + b - a b) + 4 (1 + b)
    spec = {Axis("p", 0): PermutationGroup(1, {Axis("p", 0)}, set()), Axis("q", 0): PermutationGroup(1, {Axis("q", 0)}, set())}
    toy = [(10, Axis("p", 0)), (6, Axis("p", 0), Axis("q", 0)), (4, Axis("q", 0))]
    ww = {Axis("p", 0): 1.0, Axis("q", 0): 0.8}
    grid = np.linspace(0, 1, 201)
    for budget in (1.2, 1.45, 1.7):
        r = qp_ratios(spec, toy, budget, ww)
        best = max(ww[Axis("p", 0)] * a + ww[Axis("q", 0)] * b for a in grid for b in grid
                   if partial_merge_flops(spec, toy, {Axis("p", 0): a, Axis("q", 0): b}) <= budget * 20)
        assert sum(ww[k] * r[k] for k in r) >= best - 1e-2


def test_eval_helpers_route_features_to_the_right_head(tiny_basic):
    """A merged 'backbone' that emits [merged | separate-1 | separate-2] features: with the blocks of get_fc_perm,
    eval_perm_model must reproduce each source head's own predictions (accuracy 1 against them)."""
    from pleas.methods.pleas_merging import eval_perm_model, eval_whole_model, get_fc_perm

    t = tiny_basic
    perm, costs = t.per_key("am_perm"), t.per_key("am_cost")
    b1, b2, b1c, b2c = get_fc_perm(perm, t.spec, costs, 0.5)
    width = len(b1) + len(b1c)
    g = torch.Generator().manual_seed(5)
    f1, f2 = torch.randn(16, width, generator=g), torch.randn(16, width, generator=g)
    f2[:, b2] = f1[:, b1]                                 # merged units carry one shared value
    merged_feats = torch.cat([f1[:, b1], f1[:, b1c], f2[:, b2c]], 1)

    class Backbone(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dummy = torch.nn.Parameter(torch.zeros(1))

        def forward(self, idx):
            return merged_feats[idx.long().flatten()]

    heads = [torch.nn.Linear(width, 10), torch.nn.Linear(width, 10)]
    for idx, (head, feats) in enumerate(zip(heads, (f1, f2))):
        labels = head(feats).argmax(1)
        loader = [(torch.arange(0, 8).view(8, 1), labels[:8]), (torch.arange(8, 16).view(8, 1), labels[8:])]
        assert float(eval_perm_model(Backbone(), head, loader, 10, (b1, b2, b1c, b2c), idx)) == 1.0
    whole = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4, 3))
    x = torch.randn(6, 4, generator=g)
    assert float(eval_whole_model(whole, [(x, whole(x).argmax(1))], 3)) == 1.0


def test_tap_views_and_pointer_tables():
    """Grouped source forwards hand every update a lazily sliced view of the group's taps; the per-update fast path of
    PleasFitter needs only the addresses of those slices, computed from one table per forward."""
    from pleas_merging_amd.methods.pleas_merging import TapDict, _TapView, tap_pointers

    base = TapDict(a=torch.arange(6 * 3 * 4, dtype=torch.float32).reshape(6, 3, 4),
                   b=torch.arange(6 * 5, dtype=torch.float32).reshape(6, 5))
    view = _TapView(base, 2, 4)
    assert "a" in view and "zz" not in view and len(view) == 2 and list(view.keys()) == ["a", "b"]
    assert torch.equal(view["a"], base["a"][2:4]) and dict(view.items())["b"].shape == (2, 5)
    names = ("b", "a")
    got = tap_pointers(view, names)
    assert [int(p) for p in got] == [base["b"][2:4].data_ptr(), base["a"][2:4].data_ptr()]
    assert [int(p) for p in tap_pointers(base, names)] == [base["b"].data_ptr(), base["a"].data_ptr()]
    assert base.packs and tap_pointers(_TapView(base, 4, 6), names)[1] == base["a"][4:].data_ptr()   # table reused
    # a tensor the grouped kernels could not read in place -> no table (the caller falls back to the layer-by-layer path)
    assert tap_pointers(TapDict(a=torch.zeros(4, 6).t()), ("a",)) is None


def test_get_blocks_memo_sees_in_place_changes():
    """get_blocks keeps its latest answer (partial_merge and PleasFitter ask for the same blocks back to back); the key
    holds every tensor's version counter, so a changed cost matrix or permutation is recomputed."""
    from pleas_merging_amd.core.utils import Axis, PermutationGroup
    from pleas_merging_amd.methods.partial_matching import get_blocks

    ax = Axis("w", 0)
    spec = {ax: PermutationGroup(size=4, state=[ax], node=[])}
    perm = {ax: torch.tensor([1, 0, 3, 2])}
    costs = {ax: torch.tensor([[0., 9., 0., 0.], [8., 0., 0., 0.], [0., 0., 0., 1.], [0., 0., 2., 0.]])}
    first = get_blocks(spec, perm, costs, 0.5)
    again = get_blocks(spec, perm, costs, 0.5)
    assert all(a is b for a, b in zip(first[ax], again[ax]))                    # served from the memo
    assert first[ax][0].tolist() == [0, 1] and first[ax][2].tolist() == [2, 3]
    costs[ax][2, 3] = 20.0                                                      # in place: version counter moves
    changed = get_blocks(spec, perm, costs, 0.5)
    assert changed[ax][0].tolist() == [0, 2] and changed[ax][2].tolist() == [1, 3]
    assert get_blocks(spec, perm, costs, 0.25)[ax][0].tolist() == [0, 1, 2]     # another ratio is another key


def test_bench_starts_its_own_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher: the parent builds a torch.distributed.run command for N ranks on
    127.0.0.1 and hands over its own arguments; with WORLD_SIZE set (the driver's launch) nothing is re-launched."""
    import importlib
    import subprocess
    import sys

    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    rc = bench.self_launch(bench.parse(["--gpus", "4", "--steps", "2", "--warmup", "1"]))
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # defaults finish within minutes: a step is one whole job (~6.5 s), 1 warm-up + 3 timed
    d = bench.parse([])
    assert (d.gpus, d.steps, d.warmup, d.match_batches, d.updates) == (1, 3, 1, 100, 401)


def test_drop_in_namespace_exports_the_reference_surface():
    """Every name the reference's package __init__ files export (pleas/methods/__init__.py:12-41,
    pleas/core/__init__.py:9-21) resolves under the drop-in ``pleas`` namespace, so driver imports keep working."""
    import pleas.core
    import pleas.methods

    for name in ("activation_matching", "cross_features_cdist", "cross_features_inner_product", "weight_matching",
                 "partial_merge", "get_blocks", "qp_ratios", "expand_ratios", "partial_merge_flops", "train",
                 "train_eval_linear_probe", "eval_perm_model", "eval_whole_model", "get_fc_perm"):
        assert callable(getattr(pleas.methods, name)), name
    for name in ("Axis", "PermutationGroup", "PermutationSpec", "Permutation", "apply_perm", "make_identity_perm",
                 "make_random_perm", "invert_perm", "count_linear_flops", "scipy_solve_lsa"):
        assert getattr(pleas.core, name) is not None, name
    from pleas.methods.pleas_merging import train_eval_linear_probe  # noqa: F401  (the drivers' import path)


def test_linear_probe_learns_a_separable_head():
    """train_eval_linear_probe (reference pleas_merging.py:499-570): frozen backbone, Adam + cosine schedule on a fresh
    head, the reference's logging keys; on linearly separable features the probe reaches the labels."""
    from pleas.methods import train_eval_linear_probe

    g = torch.Generator().manual_seed(0)
    backbone = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 6))
    for p in backbone.parameters():
        p.requires_grad_(False)
    truth = torch.nn.Linear(6, 3)
    xs = [torch.randn(32, 12, generator=g) for _ in range(6)]
    data = [(x, truth(backbone(x)).argmax(1)) for x in xs]
    logged = []

    class Run:
        def log(self, metrics):
            logged.append(metrics)

    before = [p.clone() for p in backbone.parameters()]
    fc = train_eval_linear_probe(backbone, data[:5], data[5:], 3, Run(), "toy", lr=5e-2, epochs=30)
    assert isinstance(fc, torch.nn.Linear) and (fc.in_features, fc.out_features) == (6, 3)
    assert all(torch.equal(a, b) for a, b in zip(before, backbone.parameters()))     # backbone untouched
    assert len(logged) == 31 and set(logged[0]) == {"toy_linear_probe_train_acc", "toy_linear_probe_train_loss", "epoch",
                                                    "toy_total_loss"}
    assert logged[-1]["toy_linear_probe_acc"] >= 0.85 and logged[-2]["toy_linear_probe_train_acc"] >= 0.9
    assert logged[-2]["toy_total_loss"] < logged[0]["toy_total_loss"]


def test_host_lap_entry_point_equals_scipy():
    """``pleas_lsap_host`` (the C-ABI's explicit host entry point; no GPU involved): identical ``col_ind``
    to scipy -- the reference's solver, solvers.py:29-31 -- on the golden cases (ties, all-equal, cdist-structured, both
    directions) and on larger random fp32 / fp64 problems."""
    from scipy.optimize import linear_sum_assignment

    from pleas.core.solvers import host_solve_lsa

    z = np.load(os.path.join(GOLDEN, "lap_small.npz"))
    for i in range(int(z["n_cases"])):
        a = torch.from_numpy(z["cost_%d" % i])
        for mx in (True, False):
            got = host_solve_lsa(a, maximize=mx)
            assert got.dtype == torch.int64 and (got.numpy() == z["col_%s_%d" % ("max" if mx else "min", i)]).all(), (i, mx)
    g = torch.Generator().manual_seed(11)
    for n, dt in ((65, torch.float32), (130, torch.float64), (257, torch.float32)):
        a = torch.randn(n, n, generator=g, dtype=dt)
        ties = torch.randint(0, 3, (n, n), generator=g).to(dt)
        for m in (a, ties):
            for mx in (True, False):
                _, want = linear_sum_assignment(m.numpy(), maximize=mx)
                assert (host_solve_lsa(m, maximize=mx).numpy() == want).all(), (n, dt, mx)
    with pytest.raises(RuntimeError):
        host_solve_lsa(torch.zeros(3, 4))


def test_config0_resnet18_weight_matching_on_the_host():
    """BASELINE.json configs[0] as stated (ResNet-18 pair weight_matching ON CPU, random-init weights, no GPU, no data;
    reference driver call run_domainnet.py:247-255): host state dicts, the library's host LAP entry point and a host
    ``cross_weights`` callable, passed EXPLICITLY through the reference's plug points -- same sweeps, permutations and
    costs as the oracle (scipy's algorithm restated).  The defaults never fall back to this; they raise on CPU tensors."""
    from pleas.core.compiler import get_permutation_spec
    from pleas.core.solvers import host_solve_lsa
    from pleas.methods.weight_matching import weight_matching
    from pleas_merging_amd import resnet as zoo

    models = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        models.append(zoo.MODELS["resnet18"](num_classes=1000))
    spec = get_permutation_spec(models[0], ((1, 3, 224, 224),))
    want_p, want_c, laps = orc.weight_matching(spec, models[0].state_dict(), models[1].state_dict(), 100, 0)

    def inner(wa, wb, axis):        # the reference's cross_features_inner_product (activation_matching.py:14-28) on the host
        return wa.movedim(axis, 0).reshape(wa.shape[axis], -1) @ wb.movedim(axis, 0).reshape(wb.shape[axis], -1).t()

    perm, costs = weight_matching(spec, models[0].state_dict(), models[1].state_dict(), max_iter=100, seed=0, verbose=False,
                                  lsa_solver=host_solve_lsa, cross_weights=inner, return_costs=True)
    assert laps >= len(spec) == 12
    for k in spec:
        assert (perm[k] == want_p[k]).all(), k
        assert torch.allclose(costs[k], want_c[k], rtol=1e-5, atol=1e-5), k
    with pytest.raises(RuntimeError):
        weight_matching(spec, models[0].state_dict(), models[1].state_dict(), max_iter=1, verbose=False)
