"""Frozen source forwards with BatchNorm / residual add / ReLU folded into one HIP pass.

The PLeaS loop runs ``model1(x); model2(x)`` under forward hooks on every Conv2d / Linear
(reference pleas/methods/pleas_merging.py:267-268; both models in ``eval()``, :352-353).  Only the
hooked layers' inputs and outputs are consumed, so everything between two hooked layers is free to
be fused.  ``fuse_bn_act`` rewrites an fx trace of the model:

    bn(x) -> relu                      =>  bn_act(x, scale, shift, None, relu=True)
    bn(x) -> (+ identity) -> relu      =>  bn_act(x, scale, shift, identity, relu=True)
    bn(x)                              =>  bn_act(x, scale, shift, None, relu=False)
    bn(x) -> relu -> MaxPool2d         =>  bn_act_maxpool(x, scale, shift, kernel, stride, padding, relu=True)
                                           (the stem: the full-resolution ReLU output is never written)

    Conv2d (k x k, dense, undilated)   =>  HipConv(conv)(x): ``hip_ops.conv2d`` -- the grouped forward's tile forms as a plain
                                           convolution, bit-for-bit repeatable (``SOURCE_CONV``)
    HipConv(conv)(x) -> bn_act(...)    =>  HipConvBnAct: ONE launch that stores the convolution's output (the hooks' tensor) and
                                           its activated image from the same registers (``SOURCE_CONV_BN``)

``scale = weight / sqrt(running_var + eps)`` and ``shift = bias - running_mean * scale`` are computed
once in fp64.  The hooked modules are the SAME objects in the rewritten GraphModule, so hooks
registered on the original model keep firing; conv outputs are never written in place.
Values differ from the vendor BN kernel by fp32 rounding only (one fma instead of sub-mul-mul-add).
"""
from __future__ import annotations

import operator
import os
from typing import Callable, Optional

import torch
import torch.fx
import torch.nn.functional as F
from torch import nn

from .. import hip_ops


# Which convolutions of the frozen source / twin forwards run on the library's own kernel instead of the vendor's:
#   "all"    (default) every dense, undilated, square Conv2d.  Two reasons.  (1) Repeatability: MIOpen's immediate-mode picks for
#            k x k layers split K with atomics on small images / batches -- the SAME forward differs from itself run to run from
#            the first 3 x 3 layer on (ResNet-101: layer4's at 128 samples, layer2's at 16, layer1's at 4), which flipped near-tie
#            assignments between two runs of one job (GPUTEST_r04); its deterministic solvers (torch.backends.cudnn.deterministic)
#            are 6-15x slower per 3 x 3 layer.  (2) Time: layer by layer in isolation the vendor's 1 x 1 GEMMs are as fast or
#            faster (profiles/r05_probe_conv_classes_128.txt), but IN THE JOB -- two models on two streams beside each other,
#            no layout transposes, one dispatch path -- the own kernel for every layer is the fastest arrangement: 6.07 s per job
#            against 6.18 s with the 1 x 1 layers on the vendor's GEMM (same box; under the split-bf16 arithmetic 4.90 against
#            5.27 s; profiles/r05_exp_source_conv.txt).
#   "kxk"    only the layers with k > 1 (the 1 x 1 layers on the vendor's GEMM: measured repeatable);  "vendor"  none (rounds 1-4).
SOURCE_CONV = os.environ.get("PLEAS_SOURCE_CONV", "all")
# "1" (default): an own convolution whose only consumer is an eval-mode BatchNorm chain takes that chain into its epilogue
# (``hip_ops.conv2d_bn_act``: same bits as the two launches, the convolution's output is not read back); "0": two launches.
SOURCE_CONV_BN = os.environ.get("PLEAS_SOURCE_CONV_BN", "1")


def own_conv_ok(mod: nn.Module, mode: Optional[str] = None) -> bool:
    """Does ``mod`` go through ``hip_ops.conv2d`` under ``mode`` (default: ``SOURCE_CONV``)?"""
    mode = SOURCE_CONV if mode is None else mode
    if mode == "vendor" or type(mod) is not nn.Conv2d:
        return False
    if not (mod.groups == 1 and mod.dilation == (1, 1) and mod.padding_mode == "zeros" and isinstance(mod.padding, tuple)
            and mod.stride[0] == mod.stride[1] and mod.padding[0] == mod.padding[1] and mod.kernel_size[0] == mod.kernel_size[1]):
        return False
    k = mod.kernel_size[0]
    return mod.weight.dtype == torch.float32 and k * k <= 64 and (k > 1 or mode == "all")


class HipConv:
    """Graph callable standing in for a frozen ``nn.Conv2d``: ``hip_ops.conv2d`` on CUDA fp32 inputs without autograd, the
    module itself otherwise.  ``fire_hooks``: the module's forward hooks are called as its own ``__call__`` would (the
    PLeaS taps sit on the ORIGINAL modules, pleas_merging.py:197-231).  k x k weights with Cin % 32 == 0 are kept
    kernel-position-major (the flat-shift tile forms); the copy follows the parameter's version counter."""

    def __init__(self, conv: nn.Conv2d, tag: str, fire_hooks: bool = True):
        self.conv, self.fire_hooks = conv, fire_hooks
        self.k, self.stride, self.pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        self.kpos = self.k > 1 and conv.weight.shape[1] % 32 == 0
        self._w = self._stamp = None
        self.__name__ = self.__qualname__ = "hip_conv_%s" % tag

    def weight(self) -> torch.Tensor:
        w = self.conv.weight
        stamp = (w.data_ptr(), w._version)
        if self._stamp != stamp:
            with torch.no_grad():
                self._w = (w.detach().permute(0, 2, 3, 1) if self.kpos else w.detach()).contiguous()
            self._stamp = stamp
        return self._w

    def __call__(self, x):
        conv = self.conv
        if (not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or not conv.weight.is_cuda
                or (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad))):
            return conv(x)                      # the module's own path, hooks included
        y = hip_ops.conv2d(x if x.is_contiguous() else x.contiguous(), self.weight(), conv.bias, self.stride, self.pad, self.kpos)
        if self.fire_hooks and conv._forward_hooks:
            for hook in list(conv._forward_hooks.values()):
                out = hook(conv, (x,), y)
                if out is not None:
                    y = out
        return y


def _bn_act(x, scale, shift, res, relu):
    return hip_ops.bn_act(x, scale, shift, res, relu)


class HipConvBnAct:
    """``bn_act(HipConv(conv)(x), scale, shift, res, relu)`` as ONE launch (``hip_ops.conv2d_bn_act``): the convolution's
    output is still written -- the module's forward hooks get it, it is a regression target of the PLeaS loop -- and the
    activated image is stored beside it from the same registers.  Anything the fused kernel does not take (CPU tensors,
    autograd, an identity that is a strided view) goes through the two calls it replaces, as does the activation after a
    hook that RETURNS another output."""

    def __init__(self, own: HipConv):
        self.own = own
        self.__name__ = self.__qualname__ = own.__name__ + "_bn_act"

    def __call__(self, x, scale, shift, res, relu):
        own, conv = self.own, self.own.conv
        if (not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or not conv.weight.is_cuda or scale.dim() != 1
                or (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad))
                or (res is not None and (not res.is_contiguous() or res.dtype != torch.float32 or res.data_ptr() % 16))):
            return _bn_act(own(x), scale, shift, res, relu)
        y, z = hip_ops.conv2d_bn_act(x if x.is_contiguous() else x.contiguous(), own.weight(), conv.bias, own.stride, own.pad,
                                     own.kpos, scale, shift, res, relu)
        if own.fire_hooks and conv._forward_hooks:
            for hook in list(conv._forward_hooks.values()):
                out = hook(conv, (x,), y)
                if out is not None:
                    y, z = out, None
        return z if z is not None else _bn_act(y, scale, shift, res, relu)


def _bn_act_pool(x, scale, shift, kernel, stride, padding, relu):
    if scale is not None and scale.dim() == 2 and scale.shape[0] > 1:
        # several batches with their own maps in one forward (BN-statistics reset): the pooled pass takes one map
        return F.max_pool2d(hip_ops.bn_act(x, scale, shift, None, relu), kernel, stride, padding)
    if scale is not None and scale.dim() == 2:
        scale, shift = scale[0], shift[0]
    return hip_ops.bn_act_maxpool(x, scale, shift, kernel, stride, padding, relu)


class BatchesPerForward:
    """How many batches lie back to back in the tensor a rewritten graph is forwarding right now (read by its train-mode
    BatchNorm folds); the caller sets ``parts`` around the call."""
    parts = 1


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _pool_window(node: torch.fx.Node, mods):
    """``(kernel, stride, padding)`` when ``node`` is an ``nn.MaxPool2d`` call that ``bn_act_maxpool`` computes exactly
    (floor mode, dilation 1, no indices, one stride and one padding for both axes); None otherwise."""
    if node.op != "call_module" or len(node.args) != 1 or node.kwargs:
        return None
    m = mods.get(node.target)
    if not isinstance(m, nn.MaxPool2d) or m.ceil_mode or m.return_indices or _pair(m.dilation) != (1, 1):
        return None
    kernel, stride, padding = _pair(m.kernel_size), _pair(m.stride if m.stride is not None else m.kernel_size), _pair(m.padding)
    if stride[0] != stride[1] or padding[0] != padding[1] or 2 * padding[0] > min(kernel):
        return None
    return kernel, stride[0], padding[0]


def _is_relu(node: torch.fx.Node, mods) -> bool:
    if node.op == "call_module":
        return isinstance(mods[node.target], nn.ReLU)
    if node.op == "call_function":
        return node.target in (F.relu, torch.relu, torch.relu_)
    if node.op == "call_method":
        return node.target in ("relu", "relu_")
    return False


def _is_add(node: torch.fx.Node) -> bool:
    if node.kwargs:
        return False
    if node.op == "call_function" and node.target in (operator.add, operator.iadd, torch.add):
        return len(node.args) == 2 and all(isinstance(a, torch.fx.Node) for a in node.args)
    if node.op == "call_method" and node.target in ("add", "add_"):
        return len(node.args) == 2 and all(isinstance(a, torch.fx.Node) for a in node.args)
    return False


def _foldable(bn: nn.Module) -> bool:
    return (isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.running_mean is not None
            and bn.running_var is not None and bn.running_mean.dtype == torch.float32)


def _foldable_train(bn: nn.Module) -> bool:
    """A BatchNorm2d that normalises with the BATCH's statistics (train mode, or no running statistics at all): still
    one affine map per channel, computed per batch on the device (``hip_ops.bn_train_fold``)."""
    if not isinstance(bn, nn.BatchNorm2d) or not (bn.training or bn.running_mean is None):
        return False
    tensors = [t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None]
    return all(t.dtype == torch.float32 for t in tensors)


def fold_bn(bn: nn.BatchNorm2d):
    """``(scale, shift)`` fp32 tensors with ``bn(x) == x * scale + shift`` in eval mode (computed in fp64)."""
    with torch.no_grad():
        var = bn.running_var.double()
        w = bn.weight.double() if bn.weight is not None else torch.ones_like(var)
        b = bn.bias.double() if bn.bias is not None else torch.zeros_like(var)
        scale64 = w / torch.sqrt(var + bn.eps)
        shift64 = b - bn.running_mean.double() * scale64
    return scale64.float().contiguous(), shift64.float().contiguous()


def _train_fold(bn: nn.BatchNorm2d, tag: str, state: Optional[BatchesPerForward] = None) -> Callable:
    """Graph callable of a train-mode BatchNorm: ``x -> (scale, shift)`` of THIS batch; the module's running statistics
    and batch counter move on exactly as in the module's own forward (``hip_ops.bn_train_fold``: one streaming pass).
    ``state.parts`` > 1: the tensor holds that many batches back to back, each folded on its own samples, in order, by
    the same launch (``[parts, C]`` maps)."""
    folder = hip_ops.BnTrainFold(bn)      # workspace / output vectors / addresses looked up once per input shape

    def fold(x):
        return folder(x, state.parts if state is not None else 1)

    fold.__name__ = fold.__qualname__ = "bn_train_fold_%s" % tag
    return fold


def fuse_bn_act(model: nn.Module, op: Callable = _bn_act, train_stats: bool = False,
                pool_op: Optional[Callable] = None, conv: Optional[str] = None) -> Optional[torch.fx.GraphModule]:
    """fx copy of ``model`` (sharing its submodules) with every eval-mode BatchNorm2d chain replaced by
    ``op(x, scale, shift, residual_or_None, relu)`` -- the HIP kernel ``hip_ops.bn_act`` unless a test
    passes its own.  Returns None when the model cannot be traced or holds nothing to fold; the caller
    then runs the model as it is (vendor kernels).

    ``train_stats=True``: BatchNorm2d modules that normalise with the BATCH's statistics (train mode -- the BN-reset pass
    after merging, run_domainnet.py:327-341) are folded as well: ``scale, shift = bn_train_fold(bn, x)`` per batch,
    then the same single ``op`` pass -- x is read twice and written once, where the vendor's train-mode BatchNorm + add +
    ReLU read or write it seven times.

    ``pool_op(x, scale, shift, kernel, stride, padding, relu)`` takes a BN -> ReLU chain whose only consumer is a max
    pooling (``hip_ops.bn_act_maxpool`` with the default ``op``; with a caller's ``op`` only when passed too).

    ``conv`` ("kxk" / "all" / "vendor", default ``SOURCE_CONV``): which Conv2d calls become ``HipConv`` calls -- only with
    the default ``op`` (a test's CPU ``op`` keeps the modules) and never for a model in train mode (autograd may be on)."""
    if pool_op is None and op is _bn_act:
        pool_op = _bn_act_pool
    try:
        graph = torch.fx.Tracer().trace(model)      # the graph alone: one code generation at the end, not two
    except Exception:  # noqa: BLE001 -- untraceable control flow: nothing to rewrite
        return None
    mods = dict(model.named_modules())
    constants = {}                                    # get_attr targets of the folded scale / shift vectors
    state = BatchesPerForward()                       # train_stats: batches per forwarded tensor, set by the caller
    folded = 0
    for node in list(graph.nodes):
        if node.op != "call_module" or len(node.args) != 1 or node.kwargs:
            continue
        bn = mods.get(node.target)
        per_batch = train_stats and _foldable_train(bn)
        if not (per_batch or _foldable(bn)):
            continue
        tag = node.name
        if not per_batch:
            scale, shift = fold_bn(bn)
            constants["_pleas_scale_%s" % tag] = scale
            constants["_pleas_shift_%s" % tag] = shift

        chain, res, relu = [node], None, False
        users = list(node.users)
        if len(users) == 1 and _is_relu(users[0], mods):
            chain.append(users[0])
            relu = True
        elif len(users) == 1 and _is_add(users[0]) and users[0].args[0] is not users[0].args[1]:
            add = users[0]
            add_users = list(add.users)
            if len(add_users) == 1 and _is_relu(add_users[0], mods):
                res = add.args[1] if add.args[0] is node else add.args[0]
                chain += [add, add_users[0]]
                relu = True
        window = None
        if pool_op is not None and relu and res is None and len(chain[-1].users) == 1:
            pool = next(iter(chain[-1].users))
            window = _pool_window(pool, mods)
            if window is not None:
                chain.append(pool)
        last = chain[-1]
        with graph.inserting_before(last):
            # explicit base names: fx would otherwise derive them from the targets character by character
            if per_batch:
                stats = graph.create_node("call_function", _train_fold(bn, tag, state), (node.args[0],), {}, name="bn_stats")
                s = graph.create_node("call_function", operator.getitem, (stats, 0), {}, name="bn_scale")
                t = graph.create_node("call_function", operator.getitem, (stats, 1), {}, name="bn_shift")
            else:
                s = graph.create_node("get_attr", "_pleas_scale_%s" % tag, (), {}, name="bn_scale")
                t = graph.create_node("get_attr", "_pleas_shift_%s" % tag, (), {}, name="bn_shift")
            if window is not None:
                fused = graph.create_node("call_function", pool_op, (node.args[0], s, t) + window + (relu,), {}, name="bn_act_pool")
            else:
                fused = graph.create_node("call_function", op, (node.args[0], s, t, res, relu), {}, name="bn_act")
        last.replace_all_uses_with(fused)
        for dead in reversed(chain):
            graph.erase_node(dead)
        folded += 1
    if op is _bn_act and not model.training:
        for node in list(graph.nodes):
            if node.op == "call_module" and len(node.args) == 1 and not node.kwargs and own_conv_ok(mods.get(node.target), conv):
                with graph.inserting_before(node):
                    own = graph.create_node("call_function", HipConv(mods[node.target], node.name), (node.args[0],), {}, name=node.name + "_hip")
                node.replace_all_uses_with(own)
                graph.erase_node(node)
                folded += 1
        if SOURCE_CONV_BN == "1":
            # an own convolution consumed by one eval-mode chain only (constants, not per-batch statistics): one launch
            for node in list(graph.nodes):
                if not (node.op == "call_function" and node.target is op and isinstance(node.args[0], torch.fx.Node)):
                    continue
                src, s = node.args[0], node.args[1]
                if not (src.op == "call_function" and isinstance(src.target, HipConv) and len(src.users) == 1
                        and isinstance(s, torch.fx.Node) and s.op == "get_attr"):
                    continue
                with graph.inserting_before(node):
                    both = graph.create_node("call_function", HipConvBnAct(src.target), (src.args[0],) + tuple(node.args[1:]), {},
                                             name=src.name + "_bn_act")
                node.replace_all_uses_with(both)
                graph.erase_node(node)
                graph.erase_node(src)
    if folded == 0:
        return None
    graph.lint()
    # the GraphModule's attributes: every submodule the graph calls (the SAME objects, so hooks on them keep firing),
    # every other attribute it reads, and the folded constants
    root = dict(constants)
    for node in graph.nodes:
        if node.op == "call_module":
            root[node.target] = mods[node.target]
        elif node.op == "get_attr" and node.target not in root:
            obj = model
            for part in node.target.split("."):
                obj = getattr(obj, part)
            root[node.target] = obj
    gm = torch.fx.GraphModule(root, graph, class_name=type(model).__name__)
    gm.train(model.training)
    gm.batches_per_forward = state      # reset_bn_stats: `gm.batches_per_forward.parts = k` around a k-batch forward
    # does every normalisation layer with batch statistics go through a per-batch fold?  (only then may batches share a forward)
    gm.all_batch_statistics_folded = not uses_batch_statistics(gm)
    return gm


def uses_batch_statistics(gm) -> bool:
    """Does ``gm``'s graph still CALL something that normalises with the statistics of the batch it is given -- a BatchNorm /
    InstanceNorm module in train mode (or without running statistics), a functional ``batch_norm`` / ``instance_norm`` with
    ``training`` / ``use_input_stats`` not provably off, or a module outside ``torch.nn`` that has such a submodule?
    Several batches may share a forward only when the answer is no (the fused chains fold train-mode BatchNorm per batch)."""
    import torch.nn.functional as F

    norm = (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm)
    stat = lambda m: isinstance(m, norm) and (m.training or getattr(m, "running_mean", None) is None)
    for n in gm.graph.nodes:
        if n.op == "call_module":
            m = gm.get_submodule(n.target)
            if stat(m) or any(stat(c) for c in m.modules()):
                return True
        elif n.op == "call_function" and n.target in (F.batch_norm, torch.batch_norm):
            training = n.kwargs.get("training", n.args[5] if len(n.args) > 5 else False)
            if training is not False:
                return True
        elif n.op == "call_function" and n.target in (F.instance_norm, torch.instance_norm):
            use_input = n.kwargs.get("use_input_stats", n.args[5] if len(n.args) > 5 else True)
            if use_input is not False:
                return True
    return False
