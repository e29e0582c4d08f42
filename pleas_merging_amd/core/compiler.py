"""Permutation-spec builder: which tensor axes of a model must be permuted together.

Drop-in for ``pleas.core.compiler.get_permutation_spec`` (reference:
pleas/core/compiler.py:786-796; propagation engine :28-752).  The north star
keeps this stage on the host ("pleas/core/compiler.py unchanged"): it is a
one-off graph analysis.  Because reference files cannot ship, this is an own
implementation with a different mechanism: the reference *executes* the traced
model once on random data inside a value interpreter; here the graph is walked
symbolically over shapes recorded by ``torch.fx`` shape propagation on the meta
device (no FLOPs, no model-sized allocation), and every op contributes
*axis links* to a union-find.  The output contract is pinned by the fixtures in
tests/golden/spec_*.json captured from the reference:

* groups that reach a graph input or the graph output are dropped
  (reference :711-729), groups with <= 1 state axis are dropped (:731),
  size-1 groups are dropped (:747-748);
* the dict key of a group is its state axis that is minimal in
  ``(state_dict order, -axis)`` (:737-739);
* dict order is the order in which groups were first touched during the walk.
"""
from __future__ import annotations

import operator
from typing import Callable, Dict, List, Sequence, Tuple, Union

import torch
import torch.fx
import torch.nn.functional as F
from torch import nn
from torch.fx.passes.shape_prop import ShapeProp

from .utils import Axis, PermutationGroup, PermutationSpec, UnionFind

InputsOrShapes = Sequence[Union[tuple, torch.Tensor]]

# An endpoint is ("in", arg_index, axis) | ("out", axis) | ("state", param_name, axis).
Link = Tuple[tuple, tuple]
_MODULE_RULES: Dict[type, Callable] = {}
_FUNCTION_RULES: Dict[object, Callable] = {}
_METHOD_RULES: Dict[str, Callable] = {}


def _rule(table, *keys):
    def deco(fn):
        for k in keys:
            table[k] = fn
        return fn

    return deco


def _same_axes(n: int, arg: int = 0) -> List[Link]:
    return [(("in", arg, i), ("out", i)) for i in range(n)]


# ----------------------------------------------------------------------------- module rules
@_rule(_MODULE_RULES, nn.Conv2d)
def _conv2d(mod: nn.Conv2d, ins, out):
    (x,) = ins
    assert len(x) == 4, "Conv2d expects NCHW"
    links = [(("in", 0, 0), ("out", 0))]
    if mod.groups == 1:
        links += [(("in", 0, 1), ("state", "weight", 1)), (("out", 1), ("state", "weight", 0))]
    else:  # depthwise: channels pass straight through
        assert mod.groups == mod.in_channels == x[1] and out[1] == x[1], "only groups in {1, in_channels}"
        links += [(("in", 0, 1), ("out", 1)), (("out", 1), ("state", "weight", 0))]
    if mod.bias is not None:
        links.append((("out", 1), ("state", "bias", 0)))
    return links


@_rule(_MODULE_RULES, nn.BatchNorm2d)
def _batchnorm2d(mod, ins, out):
    (x,) = ins
    assert len(x) == 4 and tuple(x) == tuple(out)
    links = [(("in", 0, 0), ("out", 0)), (("in", 0, 1), ("out", 1))]
    for name in ("weight", "bias", "running_mean", "running_var"):
        if getattr(mod, name, None) is not None:
            links.append((("in", 0, 1), ("state", name, 0)))
    links += [(("in", 0, 2), ("out", 2)), (("in", 0, 3), ("out", 3))]
    return links


@_rule(_MODULE_RULES, nn.ReLU, nn.GELU, nn.Identity, nn.Dropout, nn.Sigmoid, nn.SiLU, nn.Tanh, nn.Hardswish)
def _elementwise_module(mod, ins, out):
    assert tuple(ins[0]) == tuple(out)
    return _same_axes(len(out))


@_rule(_MODULE_RULES, nn.AdaptiveAvgPool2d, nn.AvgPool2d, nn.MaxPool2d)
def _pool2d(mod, ins, out):
    return [(("in", 0, 0), ("out", 0)), (("in", 0, 1), ("out", 1))]


@_rule(_MODULE_RULES, nn.Linear)
def _linear(mod, ins, out):
    (x,) = ins
    last = len(x) - 1
    links = [(("in", 0, i), ("out", i)) for i in range(last)]
    links += [(("in", 0, last), ("state", "weight", 1)), (("out", last), ("state", "weight", 0))]
    if mod.bias is not None:
        links.append((("out", last), ("state", "bias", 0)))
    return links


@_rule(_MODULE_RULES, nn.LayerNorm)
def _layernorm(mod, ins, out):
    (x,) = ins
    n, m = len(x), len(mod.normalized_shape)
    links = []
    if mod.elementwise_affine:
        links += [(("state", "weight", i), ("in", 0, n - m + i)) for i in range(m)]
        if mod.bias is not None:
            links += [(("state", "bias", i), ("in", 0, n - m + i)) for i in range(m)]
    return links + _same_axes(n)


def _flatten_links(x, start: int, end: int):
    dims = len(x)
    start, end = range(dims)[start], range(dims)[end]
    kept = [i for i in range(start, end + 1) if x[i] != 1]
    assert len(kept) == 1, "flatten may only squeeze unit axes around one real axis"
    links = [(("in", 0, i), ("out", i)) for i in range(start)]
    links.append((("in", 0, kept[0]), ("out", start)))
    links += [(("in", 0, i), ("out", i - (end - start))) for i in range(end + 1, dims)]
    return links


@_rule(_MODULE_RULES, nn.Flatten)
def _flatten_module(mod, ins, out):
    return _flatten_links(ins[0], mod.start_dim, mod.end_dim)


# ----------------------------------------------------------------------------- function rules
@_rule(_FUNCTION_RULES, operator.add, operator.sub, operator.mul, operator.truediv, torch.add, torch.sub, torch.mul)
def _binop(ins, out, args, kwargs):
    x, y = ins[0], ins[1]
    if x is None or y is None:  # tensor (op) python scalar
        t = 0 if x is not None else 1
        return _same_axes(len(out), t)
    n, m = len(x), len(y)
    npad, mpad = max(m - n, 0), max(n - m, 0)
    xs, ys = (1,) * npad + tuple(x), (1,) * mpad + tuple(y)
    links = []
    for i in range(max(n, m)):
        if xs[i] == ys[i] and i >= max(npad, mpad):
            links += [(("in", 0, i - npad), ("in", 1, i - mpad)), (("in", 0, i - npad), ("out", i))]
        elif (xs[i] > ys[i] == 1) or (i < mpad and xs[i] == ys[i] == 1):
            links.append((("in", 0, i - npad), ("out", i)))
        elif (ys[i] > xs[i] == 1) or (i < npad and xs[i] == ys[i] == 1):
            links.append((("in", 1, i - mpad), ("out", i)))
        else:
            raise AssertionError("unbroadcastable shapes %s %s" % (x, y))
    return links


@_rule(_FUNCTION_RULES, torch.flatten)
def _flatten_fn(ins, out, args, kwargs):
    start = args[1] if len(args) > 1 else kwargs.get("start_dim", 0)
    end = args[2] if len(args) > 2 else kwargs.get("end_dim", -1)
    return _flatten_links(ins[0], start, end)


@_rule(_FUNCTION_RULES, F.relu, F.gelu, torch.relu, torch.sigmoid, torch.sqrt, torch.tanh, F.silu, F.dropout)
def _elementwise_fn(ins, out, args, kwargs):
    assert tuple(ins[0]) == tuple(out)
    return _same_axes(len(out))


@_rule(_FUNCTION_RULES, F.adaptive_avg_pool2d, F.avg_pool2d, F.max_pool2d)
def _pool_fn(ins, out, args, kwargs):
    return [(("in", 0, 0), ("out", 0)), (("in", 0, 1), ("out", 1))]


@_rule(_FUNCTION_RULES, operator.matmul, torch.matmul)
def _matmul(ins, out, args, kwargs):
    x, y = ins[0], ins[1]
    assert len(x) == len(y) == 2
    return [(("in", 0, 0), ("out", 0)), (("in", 0, 1), ("in", 1, 0)), (("in", 1, 1), ("out", 1))]


@_rule(_FUNCTION_RULES, torch.permute)
def _permute_fn(ins, out, args, kwargs):
    order = args[1]
    return [(("in", 0, order[i]), ("out", i)) for i in range(len(out))]


# ----------------------------------------------------------------------------- method rules
@_rule(_METHOD_RULES, "view", "reshape")
def _reshape(ins, out, args, kwargs):
    before, after = list(ins[0]), list(out)
    a = b = 0
    abuf = bbuf = 1
    links = []
    while a < len(after) and b < len(before):
        if abuf == bbuf and after[a] == before[b]:
            links.append((("in", 0, b), ("out", a)))
            abuf = bbuf = 1
            a += 1
            b += 1
        elif abuf <= bbuf:
            abuf *= after[a]
            a += 1
        else:
            bbuf *= before[b]
            b += 1
    return links


@_rule(_METHOD_RULES, "flatten")
def _flatten_method(ins, out, args, kwargs):
    start = args[1] if len(args) > 1 else kwargs.get("start_dim", 0)
    end = args[2] if len(args) > 2 else kwargs.get("end_dim", -1)
    return _flatten_links(ins[0], start, end)


@_rule(_METHOD_RULES, "add", "sub", "mul", "add_", "sub_", "mul_")
def _binop_method(ins, out, args, kwargs):
    return _binop(ins, out, args, kwargs)


@_rule(_METHOD_RULES, "permute")
def _permute_method(ins, out, args, kwargs):
    order = args[1:] if not isinstance(args[1], (tuple, list)) else args[1]
    return [(("in", 0, order[i]), ("out", i)) for i in range(len(out))]


@_rule(_METHOD_RULES, "to", "type", "pow", "sqrt", "contiguous", "float", "relu", "relu_", "clone", "detach")
def _elementwise_method(ins, out, args, kwargs):
    assert tuple(ins[0]) == tuple(out)
    return _same_axes(len(out))


@_rule(_METHOD_RULES, "sum", "mean")
def _reduce_method(ins, out, args, kwargs):
    x = ins[0]
    n = len(x)
    if len(args) < 2 and "dim" not in kwargs:
        return []
    dims = args[1] if len(args) > 1 else kwargs["dim"]
    dims = [d % n for d in (dims if isinstance(dims, (list, tuple)) else [dims])]
    keepdim = bool(kwargs.get("keepdim") or (len(args) > 2 and args[2]))
    links, o = [], 0
    for i in range(n):
        if i in dims:
            o += 1 if keepdim else 0
        else:
            links.append((("in", 0, i), ("out", o)))
            o += 1
    return links


@_rule(_METHOD_RULES, "size", "dim")
def _no_links(ins, out, args, kwargs):
    return []


# ----------------------------------------------------------------------------- the walk
def _shape_of(node) -> Union[Tuple[int, ...], None]:
    meta = node.meta.get("tensor_meta") if isinstance(node, torch.fx.Node) else None
    return tuple(meta.shape) if meta is not None and hasattr(meta, "shape") else None


class AxisTracer:
    """Walks a shape-annotated fx graph and unions linked axes."""

    def __init__(self, gm: torch.fx.GraphModule, verbose: bool = False):
        self.gm = gm
        self.verbose = verbose
        self.uf = UnionFind()
        self.io_items: set = set()

    def _resolve(self, node: torch.fx.Node, endpoint: tuple):
        kind = endpoint[0]
        if kind == "out":
            return ("node", node.name, endpoint[1])
        if kind == "state":
            return ("state", "%s.%s" % (node.target, endpoint[1]), endpoint[2])
        tensor_args = [a for a in node.args if True]
        src = tensor_args[endpoint[1]]
        assert isinstance(src, torch.fx.Node), "link to a non-node argument"
        return ("node", src.name, endpoint[2])

    def _link(self, node, links: Sequence[Link]):
        for a, b in links:
            self.uf.union(self._resolve(node, a), self._resolve(node, b), add=True)

    def run(self) -> "AxisTracer":
        modules = dict(self.gm.named_modules())
        n_inputs = 0
        for node in self.gm.graph.nodes:
            shape = _shape_of(node)
            ins = [_shape_of(a) for a in node.args]
            if node.op == "placeholder":
                for i in range(len(shape)):
                    io = ("placeholder", n_inputs, i)
                    self.io_items.add(io)
                    self.uf.union(io, ("node", node.name, i), add=True)
                n_inputs += 1
            elif node.op == "get_attr":
                for i in range(len(shape)):
                    self.uf.union(("state", node.target, i), ("node", node.name, i), add=True)
            elif node.op == "output":
                (res,) = node.args
                assert isinstance(res, torch.fx.Node), "single tensor output expected"
                for i in range(len(_shape_of(res))):
                    io = ("result", 0, i)
                    self.io_items.add(io)
                    self.uf.union(io, ("node", res.name, i), add=True)
            elif node.op == "call_module":
                mod = modules[node.target]
                rule = _MODULE_RULES.get(type(mod))
                if rule is None:
                    raise NotImplementedError("No handler for module %r" % (mod,))
                self._link(node, rule(mod, ins, shape))
            elif node.op == "call_function":
                if not any(s is not None for s in ins):
                    continue  # does not touch a tensor
                rule = _FUNCTION_RULES.get(node.target)
                if rule is None:
                    raise NotImplementedError("No handler for function %r" % (node.target,))
                self._link(node, rule(ins, shape, node.args, node.kwargs))
            elif node.op == "call_method":
                rule = _METHOD_RULES.get(node.target)
                if rule is None:
                    raise NotImplementedError("No handler for method call %s" % (node.target,))
                self._link(node, rule(ins, shape, node.args, node.kwargs))
            if self.verbose:
                print(node.op, node.name, ins, "->", shape)
        return self

    def spec(self) -> PermutationSpec:
        state = self.gm.state_dict()
        order = {k: i for i, k in enumerate(state.keys())}
        spec: PermutationSpec = {}
        for members in self.uf.groups():
            if any(m in self.io_items for m in members):
                continue
            st = {Axis(m[1], m[2]) for m in members if m[0] == "state"}
            nd = {Axis(m[1], m[2]) for m in members if m[0] == "node"}
            if len(st) <= 1:
                continue
            sizes = {state[a.key].shape[a.axis] for a in st}
            assert len(sizes) == 1, "inconsistent sizes in one group: %s" % sorted(map(str, st))
            size = sizes.pop()
            if size == 1:
                continue
            key = min(st, key=lambda a: (order[a.key], -a.axis))
            spec[key] = PermutationGroup(int(size), st, nd)
        return spec


def trace_with_shapes(model: nn.Module, inputs_or_shapes: InputsOrShapes) -> torch.fx.GraphModule:
    """``symbolic_trace`` + shape propagation without real compute: the traced module runs ONCE on ``meta`` tensors with its
    parameters and buffers swapped for ``meta`` stand-ins for the duration of the run (no copy of the module; a deep copy
    into fake tensors, which ``ShapeProp(fake_mode=...)`` makes, was two thirds of this function's 0.5 s on ResNet-101).
    Falls back to that fake-tensor run when an operator has no ``meta`` kernel."""
    gm = torch.fx.symbolic_trace(model)
    shapes = [tuple(s) if isinstance(s, (tuple, list, torch.Size)) else tuple(s.shape) for s in inputs_or_shapes]
    was_training = gm.training
    gm.eval()  # BN in eval: the stand-in running statistics are not touched either way, and no batch-size-1 complaint
    try:
        try:
            from torch.nn.utils.stateless import _reparametrize_module

            meta = {k: torch.empty_like(v, device="meta") for k, v in list(gm.named_parameters()) + list(gm.named_buffers())}
            with torch.no_grad(), _reparametrize_module(gm, meta):
                ShapeProp(gm).propagate(*[torch.empty(s, device="meta") for s in shapes])
        except (ImportError, NotImplementedError, RuntimeError):
            from torch._subclasses.fake_tensor import FakeTensorMode

            params = list(model.parameters())
            device = params[0].device if params else torch.device("cpu")
            with FakeTensorMode(allow_non_fake_inputs=True) as mode:
                ShapeProp(gm, fake_mode=mode).propagate(*[torch.empty(s, device=device) for s in shapes])
    finally:
        gm.train(was_training)
    return gm


def get_permutation_spec(model: nn.Module, inputs_or_shapes: InputsOrShapes, verbose: bool = False) -> PermutationSpec:
    """Reference: pleas/core/compiler.py:786-796 (same signature and result)."""
    gm = trace_with_shapes(model, inputs_or_shapes)
    return AxisTracer(gm, verbose=verbose).run().spec()


@torch.no_grad()
def check_permutation_spec(model: nn.Module, spec: PermutationSpec, x: torch.Tensor, rtol=1e-2, atol=1e-3) -> bool:
    """Function-invariance self-check (reference: compiler.py:754-783): permuting
    any single group of ``model`` must leave ``model(x)`` unchanged."""
    from .utils import apply_perm

    saved = {k: v.clone() for k, v in model.state_dict().items()}
    ref = model(x)
    ok = True
    for key, group in spec.items():
        p = torch.randperm(group.size)
        apply_perm({key: p}, spec, model, inplace=True)
        ok &= bool(torch.allclose(model(x), ref, rtol=rtol, atol=atol))
        model.load_state_dict(saved)
    return ok
