"""Activation matching: per-group similarity of two models' activations + one LAP per group.

Drop-in for the reference's ``pleas/methods/activation_matching.py``
(``cross_features_*`` :14-46, ``build_cross_module`` :49-100,
``compute_matching_costs`` :103-136, ``activation_matching`` :139-177).

MI355X design.  Both models run under PyTorch-ROCm inside ONE fx graph (twin copies of every
node).  Right after each tracked node -- i.e. before a following in-place ReLU can overwrite
it -- the graph calls a *sink*.  With the default HIP cross features the sink is
``pleas_gram_accum``: an fp32-MFMA contraction that reads both NCHW activations in place and
adds ``-cdist`` (or the inner product) straight into the node's *group* matrix inside one
flat fp32 arena.  No per-node C x C tensors, no Python ``sum`` over nodes, and the arena is
what a multi-GPU run all-reduces (one RCCL call) before the batched LAP kernel solves every
group at once.  Any other ``cross_features`` / ``lsa_solver`` callable is honoured through
the reference's plug points (generic path).
"""
from __future__ import annotations

import os

from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch
import torch.fx
from torch import nn

from ..core.solvers import hip_solve_lsa
from ..core.utils import Axis, Permutation, PermutationSpec
from .. import hip_ops
from ..hip_ops import cross_features_cdist, cross_features_inner_product  # noqa: F401  (public plug-ins)

_FUSED_EPILOGUE = {}  # callable -> epilogue id, filled below


def _register_fused():
    from .. import hip_ops

    _FUSED_EPILOGUE[hip_ops.cross_features_cdist] = hip_ops.EPI_NEG_CDIST
    _FUSED_EPILOGUE[hip_ops.cross_features_inner_product] = hip_ops.EPI_INNER


_register_fused()


# ------------------------------------------------------------------------------------------ twin graph
def _out_of_place(twin: torch.fx.Graph, node: torch.fx.Node, lookup: Callable, submods) -> Optional[torch.fx.Node]:
    """Out-of-place twin of an in-place op (same values, new tensor), or None when ``node`` is not one.
    Deferred sinks read tracked activations after the forward, so nothing may overwrite them."""
    import operator

    import torch.nn.functional as F

    if node.op == "call_module" and isinstance(submods[node.target], nn.ReLU) and submods[node.target].inplace:
        return twin.call_function(torch.relu, (lookup(node.args[0]),))
    if node.op == "call_function":
        if node.target is operator.iadd:
            return twin.call_function(operator.add, tuple(torch.fx.node.map_arg(node.args, lookup)))
        if node.target is torch.relu_ or (node.target is F.relu and (node.kwargs.get("inplace") or
                                                                    (len(node.args) > 1 and node.args[1]))):
            return twin.call_function(torch.relu, (lookup(node.args[0]),))
    if node.op == "call_method" and node.target in ("relu_", "add_") and not node.kwargs:
        return twin.call_method(node.target[:-1], tuple(torch.fx.node.map_arg(node.args, lookup)))
    return None


def _own_conv(twin: torch.fx.Graph, node: torch.fx.Node, side: int, model: nn.Module, env) -> Optional[torch.fx.Node]:
    """The twin-graph call of a k x k convolution on the library's own kernel (``source_forward.HipConv``: bit-for-bit
    repeatable, which the vendor's 3 x 3 kernels are not), or None when ``node`` is not one / the mode says vendor."""
    from .source_forward import HipConv, own_conv_ok

    if node.op != "call_module" or len(node.args) != 1 or node.kwargs:
        return None
    try:
        mod = model.get_submodule(node.target)
    except AttributeError:
        return None
    if not own_conv_ok(mod):
        return None
    return _node(twin, "call_function", HipConv(mod, "%d_%s" % (side, node.name)), (env[node.args[0]],), "hip_conv")


def _node(graph: torch.fx.Graph, op: str, target, args=(), name: str = "n") -> torch.fx.Node:
    """``graph.create_node`` with an explicit (already valid) base name: without one fx derives it from the target
    through a per-character Python loop (``_snake_case``), ~35 us per node and 0.1 s per job for our ~2000 nodes."""
    return graph.create_node(op, target, tuple(args), {}, name=name)


class _Trace:
    """fx graph of a model without the GraphModule around it (``symbolic_trace`` also generates and compiles Python code
    for a module nobody calls: a third of the twin graph's build time, spent while the GPU waits for its first batch)."""

    def __init__(self, model: nn.Module):
        self.graph = torch.fx.Tracer().trace(model)
        self._model = model

    def named_modules(self):
        return self._model.named_modules()


class _SideStream:
    """Stream fork/join calls placed in a split twin graph: model2's chain is enqueued on a second HIP
    stream, model1's on the caller's stream, and the caller's stream waits for both before the sinks run.
    Side-stream tensors are consumed on the caller's stream and re-used by the side stream only after the
    next ``fork`` (which waits for everything enqueued on the caller's stream) -- no record_stream needed."""

    def __init__(self, device: torch.device):
        self.side = hip_ops.role_stream(device, "model2")
        self.main = None

        def fork(x):
            self.main = torch.cuda.current_stream(x.device)
            self.side.wait_stream(self.main)
            torch.cuda.set_stream(self.side)
            return x

        def back():
            torch.cuda.set_stream(self.main)

        def join():
            self.main.wait_stream(self.side)

        fork.__name__ = fork.__qualname__ = "pleas_stream_fork"
        back.__name__ = back.__qualname__ = "pleas_stream_back"
        join.__name__ = join.__qualname__ = "pleas_stream_join"
        self.fork, self.back, self.join = fork, back, join

    def restore(self):
        """After an exception inside the graph: make the caller's stream current again."""
        if self.main is not None:
            torch.cuda.set_stream(self.main)


def _build_twin(model1: nn.Module, model2: nn.Module, axes: Iterable[Axis], emit: Callable,
                keep_inputs: bool = False, side_stream: Optional[_SideStream] = None, fuse_bn: bool = False,
                emit_derived: Optional[Callable] = None):
    """Twin graph of ``model1``/``model2``; ``emit(graph, name, axis, node1, node2)`` adds the call
    made right after tracked node ``name`` and returns the fx node that holds its value.
    ``keep_inputs`` runs in-place activations out of place (same values, new tensor), so that tracked
    activations stay intact until the end of the forward pass.
    ``side_stream`` (needs ``keep_inputs``): instead of interleaving the two models node by node, emit model2's
    whole chain on the side stream, then model1's chain, then join and only then the ``emit`` calls."""
    if side_stream is not None:
        if not keep_inputs:
            raise ValueError("a split twin graph defers its sinks: it needs keep_inputs=True")
        return _build_split_twin(model1, model2, axes, emit, side_stream, fuse_bn, emit_derived)
    traced = _Trace(model1)
    submods = dict(traced.named_modules())
    want: Dict[str, List[int]] = {}
    for ax in axes:
        if ax.axis not in want.setdefault(ax.key, []):
            want[ax.key].append(ax.axis)

    twin = torch.fx.Graph()
    env = ({}, {})  # original node -> new node, per model
    cross: Dict[Tuple[str, int], torch.fx.Node] = {}
    result = None
    for node in traced.graph.nodes:
        if node.op == "placeholder":
            shared = twin.placeholder(node.target)
            env[0][node] = env[1][node] = shared
            continue
        if node.op == "output":
            (ret,) = node.args
            result = ([env[0][ret], env[1][ret]], cross)
            continue
        made = []
        for side in (0, 1):
            new = _out_of_place(twin, node, lambda n, side=side: env[side][n], submods) if keep_inputs else None
            if new is not None:
                env[side][node] = new
                made.append(new)
                continue
            new = _own_conv(twin, node, side, (model1, model2)[side], env[side])
            if new is None:
                new = twin.node_copy(node, lambda n, side=side: env[side][n])
                if node.op in ("call_module", "get_attr"):
                    new.target = "%d.%s" % (side, node.target)
            env[side][node] = new
            made.append(new)
        for a in want.get(node.name, ()):
            cross[node.name, a] = emit(twin, node.name, a, made[0], made[1])
    twin.output(result)
    gm = torch.fx.GraphModule(nn.ModuleList([model1, model2]), twin)
    gm.graph.lint()
    return gm


def _bn_chains(traced: _Trace, model1: nn.Module, model2: nn.Module):
    """``BatchNorm2d -> [+ other] -> [ReLU]`` chains of the traced graph whose links have no other consumer:
    ``{last node of the chain: (bn, add or None, relu or None, residual operand or None)}`` and the set of nodes that are
    produced by the chain's single fused launch instead of their own.  The BatchNorm may be in eval mode (constants
    folded once) or in train mode (batch statistics folded per batch, ``hip_ops.bn_train_fold``), per model."""
    from .source_forward import _foldable, _foldable_train, _is_add, _is_relu

    mods1, mods2 = dict(model1.named_modules()), dict(model2.named_modules())
    fx_mods = dict(traced.named_modules())
    at, absorbed, claimed = {}, set(), set()
    for node in traced.graph.nodes:
        if node.op != "call_module" or len(node.args) != 1 or node.kwargs:
            continue
        if not all(_foldable(m) or _foldable_train(m) for m in (mods1.get(node.target), mods2.get(node.target))):
            continue
        add = relu = res = None
        users = list(node.users)
        if len(users) == 1 and _is_relu(users[0], fx_mods):
            relu = users[0]
        elif len(users) == 1 and _is_add(users[0]) and users[0] not in claimed and users[0].args[0] is not users[0].args[1]:
            add = users[0]
            claimed.add(add)
            res = add.args[1] if add.args[0] is node else add.args[0]
            after = list(add.users)
            if len(after) == 1 and _is_relu(after[0], fx_mods):
                relu = after[0]
        last = relu or add or node
        at[last] = (node, add, relu, res)
        absorbed |= {n for n in (node, add, relu) if n is not None and n is not last}
    return at, absorbed


def _train_fold(mod: nn.BatchNorm2d, side: int, name: str, streams=None) -> Callable:
    """Graph callable of a train-mode BatchNorm: ``x -> (scale, shift)`` of THIS batch (and the module's running
    statistics move on, as in the module's own forward).  When the forward carries several batches back to back
    (``streams.parts`` > 1) every batch is folded on its own samples, in order, by the same launch: ``[parts, C]`` maps."""
    folder = hip_ops.BnTrainFold(mod)     # workspace / output vectors / addresses looked up once per input shape

    def fold(x):
        return folder(x, getattr(streams, "parts", 1))

    fold.__name__ = fold.__qualname__ = "bn_train_fold_%d_%s" % (side, name)
    return fold


def _build_split_twin(model1: nn.Module, model2: nn.Module, axes: Iterable[Axis], emit: Callable, streams: _SideStream,
                      fuse_bn: bool = False, emit_derived: Optional[Callable] = None):
    import operator

    from .. import hip_ops
    from .source_forward import _foldable, fold_bn

    traced = _Trace(model1)
    submods = dict(traced.named_modules())
    root = nn.ModuleList([model1, model2])
    chains, absorbed = _bn_chains(traced, model1, model2) if fuse_bn else ({}, set())
    want_early: Dict[str, List[int]] = {}
    for ax in axes:
        want_early.setdefault(ax.key, []).append(ax.axis)
    # BatchNorm nodes of fused chains that are tracked on the channel axis together with their input (the convolution
    # output): their matching cost is derived from the input node's contraction instead of contracting again
    derivable = set()
    if emit_derived is not None:
        for bn, _add, _relu, _res in chains.values():
            src = bn.args[0]
            if isinstance(src, torch.fx.Node) and src.op not in ("placeholder", "output") and src not in absorbed \
                    and want_early.get(bn.name) == [1] and 1 in want_early.get(src.name, ()):
                derivable.add(bn)
    fold_attrs: Tuple[dict, dict] = ({}, {})
    want: Dict[str, List[int]] = {}
    for ax in axes:
        if ax.axis not in want.setdefault(ax.key, []):
            want[ax.key].append(ax.axis)
    twin = torch.fx.Graph()
    env = ({}, {})
    shared = {}
    for node in traced.graph.nodes:
        if node.op == "placeholder":
            shared[node] = twin.placeholder(node.target)
    ret = None
    for side in (1, 0):           # model2 first: its kernels are already queued on the side stream while model1 is enqueued
        first = True
        for node in traced.graph.nodes:
            if node.op == "placeholder":
                env[side][node] = shared[node]
                if side == 1 and first:   # fork on the first input: the side stream waits for it, then becomes current
                    env[side][node] = _node(twin, "call_function", streams.fork, (shared[node],), "fork")
                    first = False
                continue
            if node.op == "output":
                (ret,) = node.args
                continue
            if node in chains:
                # eval-mode BatchNorm -> [+ residual] -> [ReLU] whose links feed nothing else: ONE launch produces every
                # node of the chain (activation matching measures all of them), reading x and the residual once
                bn, add, relu, res = chains[node]
                mod = (model1, model2)[side].get_submodule(bn.target)
                if _foldable(mod):          # eval mode: scale / shift are constants of the run
                    names = ["_pleas_%s_%d_%s" % (kind, side, bn.name) for kind in ("scale", "shift")]
                    for nm_, buf in zip(names, fold_bn(mod)):
                        root.register_buffer(nm_, buf, persistent=False)
                    attrs = (_node(twin, "get_attr", names[0], (), "bn_scale"), _node(twin, "get_attr", names[1], (), "bn_shift"))
                else:                       # train mode (the reference drivers' mode): this batch's statistics, one pass
                    fold = _node(twin, "call_function", _train_fold(mod, side, bn.name, streams), (env[side][bn.args[0]],), "bn_stats")
                    attrs = (_node(twin, "call_function", operator.getitem, (fold, 0), "bn_scale"),
                             _node(twin, "call_function", operator.getitem, (fold, 1), "bn_shift"))
                fold_attrs[side][bn] = attrs
                fused = _node(twin, "call_function", hip_ops.bn_act_tracked,
                              (env[side][bn.args[0]], attrs[0], attrs[1], env[side][res] if res is not None else None,
                               relu is not None, bn not in derivable), "bn_chain")
                for slot, member in enumerate((bn, add, relu)):
                    if member is not None:
                        env[side][member] = _node(twin, "call_function", operator.getitem, (fused, slot), "chain_out")
                continue
            if node in absorbed:
                continue
            new = _out_of_place(twin, node, lambda n, side=side: env[side][n], submods)
            if new is not None:
                env[side][node] = new
                continue
            own = _own_conv(twin, node, side, (model1, model2)[side], env[side])
            if own is not None:
                env[side][node] = own
                continue
            new = twin.node_copy(node, lambda n, side=side: env[side][n])
            if node.op in ("call_module", "get_attr"):
                new.target = "%d.%s" % (side, node.target)
            env[side][node] = new
        if side == 1:
            _node(twin, "call_function", streams.back, (), "back")
    _node(twin, "call_function", streams.join, (), "join")
    cross: Dict[Tuple[str, int], torch.fx.Node] = {}
    for node in traced.graph.nodes:
        for a in want.get(node.name, ()) if node.op not in ("placeholder", "output") else ():
            if node in derivable:
                cross[node.name, a] = emit_derived(twin, node.name, node.args[0].name, a, fold_attrs[0][node], fold_attrs[1][node])
            else:
                cross[node.name, a] = emit(twin, node.name, a, env[0][node], env[1][node])
    twin.output(([env[0][ret], env[1][ret]], cross))
    gm = torch.fx.GraphModule(root, twin)
    gm.graph.lint()
    return gm


def build_cross_module(model1: nn.Module, model2: nn.Module, axes: Iterable[Axis], cross_features: Callable):
    """One GraphModule that runs ``model1`` and ``model2`` side by side on the same input and
    calls ``cross_features(act1, act2, axis)`` right after every tracked node.

    Same contract as the reference (:49-100): ``model1`` is traced and its graph reused for
    ``model2`` (isomorphic models), submodule targets are prefixed ``0.`` / ``1.`` under a
    ``ModuleList([model1, model2])`` root, and the module returns
    ``([out1, out2], {(node_name, axis): cross_value})``.
    """
    return _build_twin(model1, model2, axes,
                       lambda g, name, a, n1, n2: g.call_function(cross_features, (n1, n2, a)))


class GroupArena:
    """Flat fp32 buffer holding every group's C x C cost matrix back to back (one all-reduce)."""

    def __init__(self, spec: PermutationSpec, device: torch.device):
        self.keys = list(spec.keys())
        sizes = [spec[k].size for k in self.keys]
        self.flat = torch.zeros(sum(s * s for s in sizes), dtype=torch.float32, device=device)
        self.view: Dict[Axis, torch.Tensor] = {}
        off = 0
        for k, s in zip(self.keys, sizes):
            self.view[k] = self.flat[off:off + s * s].view(s, s)
            off += s * s

    def zero_(self):
        self.flat.zero_()


class _FusedSink:
    """Callables placed in the twin graph.  ``grouped=True`` (default): a sink only hands the node's
    two activations to a :class:`GramBatch`; ONE grouped launch per batch contracts all nodes.
    ``grouped=False``: every sink launches ``pleas_gram_accum`` into its group matrix right away."""

    def __init__(self, arena: GroupArena, node_group: Dict[Axis, Axis], epilogue: int, grouped: bool):
        from .. import hip_ops

        self._accum = hip_ops.gram_accum
        self.arena, self.node_group, self.epilogue = arena, node_group, epilogue
        self.group_index = {key: i for i, key in enumerate(arena.keys)}
        self.index: Dict[Tuple[str, int], List[int]] = {}   # (node name, axis) -> positions in the current GramBatch
        self.batch = hip_ops.GramBatch([arena.view[k] for k in arena.keys], epilogue) if grouped else None
        # batches in the forward that is running: its input is that many batches back to back along dim 0, and every
        # tracked node is queued once PER BATCH (sample slices: contiguous slabs) -- the distance epilogue is per batch
        self.parts = 1

    def bind(self, node_name: str):
        if self.batch is not None:
            def sink(x, y, a, _name=node_name):
                group = self.group_index[self.node_group[Axis(_name, a)]]
                if self.parts == 1:
                    self.index[_name, a] = [self.batch.add(x, y, a, group)]
                else:
                    if a % x.dim() == 0 or x.shape[0] % self.parts:
                        raise RuntimeError("node %s: cannot split %s into %d batches along dim 0" % (_name, tuple(x.shape), self.parts))
                    n = x.shape[0] // self.parts
                    self.index[_name, a] = [self.batch.add(x[g * n:(g + 1) * n], y[g * n:(g + 1) * n], a, group)
                                            for g in range(self.parts)]
                return None
        else:
            def sink(x, y, a, _name=node_name):
                self._accum(x, y, a, self.arena.view[self.node_group[Axis(_name, a)]], self.epilogue, True)
                return None

        sink.__name__ = sink.__qualname__ = "gram_sink_%s" % node_name
        return sink

    def bind_derived(self, node_name: str, source_name: str):
        """Sink of an eval-mode BatchNorm node whose input ``source_name`` is tracked on the same axis: nothing is
        contracted, the group's reduce pass derives the node from the source's products, norms and row sums."""
        def sink(scale1, shift1, scale2, shift2, a, _name=node_name, _src=source_name):
            pick = lambda v, g: v[g] if v.dim() == 2 else v      # train mode: one affine map per batch of the forward
            for g, src in enumerate(self.index[_src, a]):      # one derived node per batch of the forward, like its source
                self.batch.add_derived(src, pick(scale1, g), pick(shift1, g), pick(scale2, g), pick(shift2, g),
                                       self.group_index[self.node_group[Axis(_name, a)]])
            return None

        sink.__name__ = sink.__qualname__ = "gram_derived_sink_%s" % node_name
        return sink


def build_fused_module(spec: PermutationSpec, model1: nn.Module, model2: nn.Module, arena: GroupArena, epilogue: int,
                       grouped: bool = True, overlap: bool = False, fuse_bn: bool = False, derive_bn: bool = False):
    """Twin graph whose sinks feed the group arena (the HIP fast path).  Returns (module, sinks).
    ``overlap`` (grouped only): model2's forward runs on a second HIP stream next to model1's."""
    node_group = {nax: key for key, group in spec.items() for nax in group.node}
    sinks = _FusedSink(arena, node_group, epilogue, grouped)
    sinks.streams = _SideStream(arena.flat.device) if (overlap and grouped) else None
    gm = _build_twin(model1, model2, list(node_group.keys()),
                     lambda g, name, a, n1, n2: _node(g, "call_function", sinks.bind(name), (n1, n2, a), "sink"),
                     keep_inputs=grouped,
                     side_stream=sinks.streams, fuse_bn=fuse_bn and sinks.streams is not None,
                     emit_derived=(lambda g, name, src, a, f1, f2: _node(g, "call_function", sinks.bind_derived(name, src),
                                                                         (f1[0], f1[1], f2[0], f2[1], a), "derived_sink"))
                     if (derive_bn and fuse_bn and sinks.streams is not None) else None)
    return gm, sinks


# ------------------------------------------------------------------------------------------ cost accumulation
def _dist_info():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    # PLEAS_EMULATE_WORLD=N (profiling aid, bench.py --emulate-world): this process does rank 0's share of an N-rank job
    # with the collectives skipped -- the per-rank critical path of a multi-GPU run, timed on one GPU.  The RESULTS of such
    # a run are partial sums and must not be used.
    emulated = int(os.environ.get("PLEAS_EMULATE_WORLD", "0") or 0)
    return (0, emulated) if emulated > 1 else (0, 1)


def _collectives_on() -> bool:
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def _force_collectives() -> bool:
    """PLEAS_FORCE_COLLECTIVES=1 with ``torch.distributed`` initialised: the path's exchange steps are issued even when
    the group has ONE rank (where a sum over ranks is the identity), so that the very calls an N-rank job makes --
    ``all_reduce`` of the cost arena and of the gradient arena, ``reduce_scatter_tensor`` / ``all_gather_into_tensor``
    under ``shard_optimizer`` -- go through RCCL on a one-GPU box and can be timed there."""
    return os.environ.get("PLEAS_FORCE_COLLECTIVES", "0") not in ("", "0") and _collectives_on()


def _unfolded_batch_statistics(gm: nn.Module) -> bool:
    """Does the twin graph still CALL something that uses batch statistics?  The fused chains fold a train-mode BatchNorm2d
    per batch (``pleas_bn_train_fold_batches``: exact on a concatenated forward); a module -- or a functional
    ``F.batch_norm(training=True)`` / ``F.instance_norm`` -- that is called as it is would normalise the concatenated batch
    as one, so batches then go one per forward."""
    from .source_forward import uses_batch_statistics

    return uses_batch_statistics(gm)


def _model_device(model: nn.Module) -> torch.device:
    return next(iter(model.parameters())).device


def shard_batches(dataloader, num_batches: int, rank: int, world: int):
    """Batches ``b`` with ``b % world == rank`` among the first ``num_batches`` of ``dataloader``
    (whole batches only: the distance epilogue is per batch).  Consumes the loader
    exactly like the reference's ``zip(dataloader, trange(num_batches))`` (:120)."""
    for b, (batch, _) in enumerate(zip(dataloader, range(num_batches))):
        if b % world == rank:
            yield batch


def allreduce_sum_(flat: torch.Tensor, world: int) -> torch.Tensor:
    """The one exchange step of the matching path: sum the flat cost arena over ranks
    (RCCL over xGMI under the ``nccl`` backend; gloo in the CPU tests)."""
    if (world > 1 and _collectives_on()) or _force_collectives():
        import torch.distributed as dist

        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def compute_matching_costs(spec: PermutationSpec, gm_cross: nn.Module, dataloader, num_batches: int,
                           accumulate=True, device: Optional[torch.device] = None,
                           shard: bool = True) -> Dict[Axis, torch.Tensor]:
    """Generic path over a module built by :func:`build_cross_module` with ANY ``cross_features``
    callable (reference: :103-136).  ``accumulate="reference"`` keeps only the last processed
    batch, which is what the reference computes (its membership test at :123-127 compares a
    tuple with ``Axis`` keys and never succeeds); ``True`` sums over batches (and shards them
    over ranks when ``torch.distributed`` is initialised)."""
    if device is None:
        device = _model_device(gm_cross)
    rank, world = _dist_info() if (shard and accumulate is True) else (0, 1)
    per_node: Dict[Axis, torch.Tensor] = {}
    with torch.inference_mode():
        for x, _ in shard_batches(dataloader, num_batches, rank, world):
            _, cross = gm_cross(x.to(device))
            for (name, a), v in cross.items():
                ax = Axis(name, a)
                if accumulate is True and ax in per_node:
                    per_node[ax].add_(v)
                else:
                    per_node[ax] = v
    costs = {}
    for key, group in spec.items():
        total = 0
        for nax in group.node:
            if nax in per_node:
                total = total + per_node[nax]
        if world > 1:
            if not torch.is_tensor(total):  # this rank saw no batch
                total = torch.zeros(group.size, group.size, device=device)
            allreduce_sum_(total, world)
        costs[key] = total
    return costs


def accumulate_costs_fused(spec: PermutationSpec, model1: nn.Module, model2: nn.Module, dataloader, num_batches: int,
                           epilogue: int, accumulate=True, shard: bool = True,
                           grouped: bool = True, overlap: bool = True, fuse_bn: bool = True, derive_bn: bool = True,
                           presharded: bool = False, batches_per_forward: Optional[int] = None) -> Dict[Axis, torch.Tensor]:
    """HIP fast path: every tracked node adds into its group matrix while the forwards run.

    Data parallel: with ``torch.distributed`` initialised (one process per GPU, RCCL), rank r
    takes batches ``b % world == r`` and the flat arena is all-reduced once at the end.  The
    distance epilogue is applied per batch, so sharding at batch granularity is exact
    (SURVEY.md F3).  ``accumulate="reference"`` (last batch only) does not shard.

    ``fuse_bn=True`` (default, two-stream twin only): every eval-mode ``BatchNorm2d -> [+identity] -> [ReLU]`` chain of
    the twin forward is ONE ``pleas_bn_act_tracked`` launch that keeps all nodes of the chain (680 -> 416 launches per
    batch; same folded BatchNorm as the PLeaS phase uses).  Measured on the ResNet-101 job: 17.9 -> 17.1 ms per batch.
    ``fuse_bn=False`` runs the vendor modules.

    ``derive_bn=True`` (default, needs ``fuse_bn``): a tracked eval-mode BatchNorm whose input (the convolution output) is
    tracked on the same channel axis is NOT contracted.  With ``bn(x)_i = a_i x_i + b_i`` its products and distances
    follow from the convolution node's ``G = sum x_i y_j``, squared norms and row sums
    (``<bn x_i, bn' y_j> = a a' G + a b' Sx_i + b a' Sy_j + K b b'``), which the group's reduce pass evaluates in fp64:
    104 of ResNet-101's 344 tracked nodes, i.e. ~30 % of the contraction's flops, and the BatchNorm tensors are not
    written at all.

    ``batches_per_forward`` (default: forwards of up to 64 samples, at most 4 batches): that many consecutive batches of equal shape go through the twin forward as ONE
    forward of the concatenated batch -- the vendor convolutions run 10-25 % faster per sample at 32-128 samples than at
    16 -- while every tracked node is still contracted PER BATCH (sample slices of its activations; the distance epilogue
    is per batch), all batches of the forward in one grouped launch.  Train-mode models (the reference drivers'
    mode) take part: the fused chains fold a train-mode BatchNorm2d on every batch's own samples, in order, in ONE
    launch (``pleas_bn_train_fold_batches``; statistics, running statistics and counter as after k separate forwards) and
    apply each batch's map to its sample range (``pleas_bn_act_tracked_batches``).  One batch per forward remains when a
    normalisation module with batch statistics is still CALLED by the twin graph (``fuse_bn=False``, or a module the
    chains do not cover), and when costs are not summed over batches.

    ``presharded=True`` (data parallel): ``dataloader`` already yields THIS rank's batches only (a loader over a
    ``DistributedSampler``-style partition) and ``num_batches`` counts them; nothing is skipped here and the arena is still
    all-reduced over the whole group.  The default lets every rank walk the same loader and keep every ``world``-th batch,
    which preserves the reference's consumption order (:120) but makes each rank decode all batches.
    """
    device = _model_device(model1)
    if device.type != "cuda":
        raise RuntimeError("activation_matching: models must be on the GPU for the HIP path (got %s)" % device)
    arena = GroupArena(spec, device)
    gm, sinks = build_fused_module(spec, model1, model2, arena, epilogue, grouped,
                                   overlap=overlap and grouped, fuse_bn=fuse_bn, derive_bn=derive_bn)
    rank, world = _dist_info() if (shard and accumulate is True) else (0, 1)
    take = (0, 1) if presharded else (rank, world)      # which of the loader's batches this rank contracts
    # The batches run on a created stream, not on the caller's: work on torch's default (null) stream overlaps the side
    # stream of the split twin graph markedly worse than work on a created stream (see PleasFitter).
    caller = torch.cuda.current_stream(device)
    work = hip_ops.role_stream(device, "model1") if sinks.streams is not None else caller
    work.wait_stream(caller)
    # default: forwards of up to 64 samples, at most 4 batches (the gain is the vendor convolutions' at small batches; the
    # tracked activations of a forward stay alive until its contraction) -- fixed when the first batch shows its size
    per_forward = None if batches_per_forward is None else max(1, int(batches_per_forward))
    single = accumulate is not True or sinks.batch is None or _unfolded_batch_statistics(gm)

    def forward(xs):
        if accumulate is not True:
            arena.zero_()
        sinks.parts = len(xs)
        if sinks.streams is not None:
            sinks.streams.parts = len(xs)       # read by the train-mode BatchNorm folds of the twin graph
        x = xs[0] if len(xs) == 1 else torch.cat(xs, 0)
        try:
            gm(x)
        except BaseException:
            if sinks.streams is not None:
                sinks.streams.restore()
            raise
        finally:
            sinks.parts = 1
            if sinks.streams is not None:
                sinks.streams.parts = 1
        if sinks.batch is not None:
            sinks.batch.flush(accumulate=True)

    with torch.inference_mode(), torch.cuda.stream(work):
        run: List[torch.Tensor] = []
        for x, _ in shard_batches(dataloader, num_batches, *take):
            x = hip_ops.to_device_async(x, device)      # pinned host batches: copied on a copy stream, beside the previous forward
            if single:
                per_forward = 1
            elif per_forward is None:
                per_forward = max(1, min(4, 64 // max(1, int(x.shape[0]))))
            if run and (x.shape != run[0].shape or len(run) == per_forward):
                forward(run)
                run = []
            run.append(x)
        if run:
            forward(run)
    caller.wait_stream(work)
    allreduce_sum_(arena.flat, world)
    return dict(arena.view)


def solve_all(costs: Dict[Axis, torch.Tensor], lsa_solver: Callable, while_solving: Optional[Callable] = None) -> Permutation:
    """One LAP per group.  The default solver runs all groups in ONE batched kernel launch; ``while_solving()`` (if given)
    is called after that launch is enqueued and before its results are awaited -- the largest group keeps two CUs busy
    for ~0.25 s, time the host can spend on work that does not need the permutation."""
    if lsa_solver is hip_solve_lsa:
        from .. import hip_ops

        mats = list(costs.values())
        dev = mats[0].device if mats else None
        if dev is None or dev.type != "cuda" or while_solving is None:
            outs = hip_ops.solve_lsa_batched(mats, maximize=True)
        else:
            # the kernel gets a stream of its own: what `while_solving` enqueues elsewhere (source forwards on side
            # streams) then runs beside it also when the caller is on torch's default stream
            caller, solver = torch.cuda.current_stream(dev), hip_ops.role_stream(dev, "lap")
            solver.wait_stream(caller)
            with torch.cuda.stream(solver):
                outs = hip_ops.solve_lsa_batched(mats, maximize=True)
            while_solving()
            caller.wait_stream(solver)
        return {k: o.cpu() for k, o in zip(costs.keys(), outs)}
    if while_solving is not None:
        while_solving()
    return {k: lsa_solver(v) for k, v in costs.items()}


def activation_matching(spec: PermutationSpec, model1: nn.Module, model2: nn.Module, dataloader, num_batches=1000,
                        cross_features=cross_features_cdist, lsa_solver=hip_solve_lsa, output_costs=False,
                        accumulate=True, grouped=True, while_solving: Optional[Callable] = None, presharded: bool = False,
                        batches_per_forward: Optional[int] = None):
    """Permutation of ``model2``'s units that best matches ``model1``'s activations.

    Reference: :139-177 (same positional arguments; additions: ``accumulate`` -- ``"reference"``
    reproduces the shipped last-batch-only costs -- and ``grouped`` -- one contraction launch per
    batch (default) instead of one per tracked node -- ``presharded``: under data parallelism the loader already
    yields this rank's batches only -- and ``batches_per_forward``: batches per twin forward, see
    :func:`accumulate_costs_fused`).  Returns ``perm``
    (CPU int64 per group) or ``(perm, costs)`` with fp32 costs on the compute device.
    Does not change the models' train/eval mode and does not move them.
    """
    epilogue = _FUSED_EPILOGUE.get(cross_features)
    if epilogue is not None:
        costs = accumulate_costs_fused(spec, model1, model2, dataloader, num_batches, epilogue, accumulate,
                                       grouped=grouped, presharded=presharded, batches_per_forward=batches_per_forward)
    else:
        axes = [ax for group in spec.values() for ax in group.node]
        gm = build_cross_module(model1, model2, axes, cross_features)
        costs = compute_matching_costs(spec, gm, dataloader, num_batches, accumulate, _model_device(model1))
    perm = solve_all(costs, lsa_solver, while_solving)
    return (perm, costs) if output_costs else perm
