"""PLeaS: layer-wise least-squares fitting of the partially merged model.

Drop-in for the hot-path part of the reference's ``pleas/methods/pleas_merging.py``
(``get_gradient_mask`` :11-60, ``get_model_orig_activations`` :63-149, ``capture_inputs``
:197-231, ``step`` :234-302, ``train`` :305-405).

Objective (reference :281-284): for every Conv2d/Linear layer L of the merged model,
``mean((L(ip) - op)^2)`` where ``ip`` / ``op`` are the block-merged input / output activations
of the two source layers.  ``solver="adam"`` reproduces the reference's optimiser exactly
(Adam, lr 5e-4, cosine schedule, masked gradients, MAX_STEPS+1 updates); ``solver="normal_eq"``
minimises the same objective in closed form (normal equations + Cholesky), which is what
the Adam loop approximates.

MI355X design of one Adam step (13 ms for a ResNet-101 pair at batch 16)
  * the two frozen source forwards run once each, on two HIP streams side by side, under hooks that keep every
    layer's input AND output (the reference re-runs every source layer a second time, :113-114); between the hooked
    layers BatchNorm + residual add + ReLU are one ``pleas_bn_act`` pass (``source_forward.py``);
  * the merged inputs ``ip`` of ALL layers are assembled by ONE grouped launch (``pleas_merge_batch``; reference: 8
    ``index_select`` + 2 ``cat`` per layer);
  * forward, regression target, residual and loss of ALL merged layers are ONE grouped fp32-MFMA launch
    (``pleas_fwd_batch``): the target ``op`` is gathered / averaged from the source outputs inside its epilogue and
    never materialised, only the scaled residual ``2 (out - op) / numel`` and the loss partials are written;
  * the weight gradients of ALL merged layers are ONE grouped fp32-MFMA launch (``pleas_wgrad_batch``) that reads
    residuals and inputs in place from NCHW (autograd in the reference also back-propagates into both source layers);
  * all parameters, gradients, masks and Adam moments live in flat fp32 arenas (k x k weights kernel-position-major),
    so the masked Adam update of the whole model is ONE ``pleas_masked_adam`` launch and the data-parallel gradient
    exchange is one all-reduce.
"""
from __future__ import annotations

import collections
import contextlib
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ..core.utils import Axis, get_attr
from .partial_matching import block_maps, get_blocks, spread_blocks


# ------------------------------------------------------------------------------------------ reference-shaped helpers
def get_gradient_mask(perm_blocks, model_weights: Dict[str, nn.Module]) -> List[torch.Tensor]:
    """One 0/1 mask per parameter of every layer in ``model_weights`` (reference :11-60).

    For >=2-D parameters two blocks are frozen.  The reference indexes them as
    ``mask[input_separate_1, output_separate_2] = 0`` and ``mask[input_separate_2,
    output_separate_1] = 0`` -- input slices on axis 0, output slices on axis 1 (:57-58) --
    and parity requires exactly that.  Axes missing from the spec fall back to 3 merged
    inputs / 1000 merged outputs (:31-37), i.e. no frozen block.
    """
    masks = []
    for name, layer in model_weights.items():
        frozen = _frozen_blocks(perm_blocks, name)
        for p in layer.parameters():
            mask = torch.ones_like(p)
            if p.dim() >= 2:
                for rows, cols in frozen:
                    mask[rows, cols] = 0.0
            masks.append(mask)
    return masks


def _frozen_blocks(perm_blocks, name: str):
    """The two (axis-0 slice, axis-1 slice) blocks of ``name.weight`` that stay frozen (reference :31-37, :57-58)."""
    bi = perm_blocks.get(Axis("%s.weight" % name, 1))
    bo = perm_blocks.get(Axis("%s.weight" % name, 0))
    ni, mi = (len(bi[0]), len(bi[2])) if bi is not None else (3, 0)
    no, mo = (len(bo[0]), len(bo[2])) if bo is not None else (1000, 0)
    return ((slice(ni, ni + mi), slice(no + mo, no + 2 * mo)), (slice(ni + mi, ni + 2 * mi), slice(no, no + mo)))


def _square_conv(mod: nn.Conv2d) -> bool:
    """Geometry the grouped HIP kernels take: dense, undilated, square kernel / stride / padding."""
    return (mod.groups == 1 and mod.dilation == (1, 1) and mod.stride[0] == mod.stride[1]
            and mod.padding[0] == mod.padding[1] and mod.kernel_size[0] == mod.kernel_size[1])


def prepare_sources(model1: nn.Module, model2: nn.Module):
    """The two frozen sources as the fitter runs them (eval mode, BatchNorm / add / ReLU chains folded into one HIP pass
    each; a model that cannot be rewritten is returned as it is).  Needs neither the permutation nor the merged model,
    so it can be built early and handed to ``PleasFitter(fused_sources=...)``."""
    from .source_forward import fuse_bn_act

    model1.eval()
    model2.eval()
    return fuse_bn_act(model1) or model1, fuse_bn_act(model2) or model2


class TapDict(dict):
    """name -> tensor of one forward.  ``packs`` caches, per list of names, the device addresses and per-sample strides of
    those tensors (``tap_pointers``)."""

    __slots__ = ("packs",)

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.packs = {}


class _TapView:
    """The taps of ONE update inside a grouped source forward: samples ``lo:hi`` of every tensor of ``base``, sliced on
    access (an update touches ~420 of them; slicing all of them ahead of time was 0.6 ms of host time per update)."""

    __slots__ = ("base", "lo", "hi")

    def __init__(self, base: TapDict, lo: int, hi: int):
        self.base, self.lo, self.hi = base, lo, hi

    def __getitem__(self, name):
        return self.base[name][self.lo:self.hi]

    def __contains__(self, name):
        return name in self.base

    def __len__(self):
        return len(self.base)

    def keys(self):
        return self.base.keys()

    def items(self):
        return ((k, v[self.lo:self.hi]) for k, v in self.base.items())


def tap_pointers(taps, names: tuple):
    """Device addresses (numpy uint64) of ``taps[name]`` for ``name`` in ``names`` -- of the update's first sample when
    ``taps`` is a view into a grouped forward.  The per-forward part is computed once per (forward, names)."""
    import numpy as np

    base, lo = (taps.base, taps.lo) if isinstance(taps, _TapView) else (taps, 0)
    packs = getattr(base, "packs", None)
    hit = packs.get(id(names)) if packs is not None else None
    if hit is None or hit[0] is not names:
        ts = [base[k] for k in names]
        if not all(t.is_contiguous() and t.dtype == torch.float32 for t in ts):
            return None
        hit = (names, np.array([t.data_ptr() for t in ts], dtype=np.uint64),
               np.array([t.stride(0) * 4 if t.dim() > 0 else 0 for t in ts], dtype=np.uint64))
        if packs is not None:
            packs[id(names)] = hit
    return hit[1] + np.uint64(lo) * hit[2] if lo else hit[1]


class ActivationTap:
    """Forward hooks on every Conv2d / Linear / LayerNorm of a model that keep the module's
    input and output of the latest forward (reference keeps inputs only, :197-231)."""

    KINDS = (nn.Conv2d, nn.Linear, nn.LayerNorm)

    def __init__(self, model: nn.Module):
        self.inputs: Dict[str, torch.Tensor] = TapDict()
        self.outputs: Dict[str, torch.Tensor] = TapDict()
        self.handles = []
        for name, mod in model.named_modules():
            if isinstance(mod, self.KINDS):
                self.handles.append(mod.register_forward_hook(self._hook(name)))

    def _hook(self, name):
        def hook(module, inp, out):
            if isinstance(inp, tuple):
                assert len(inp) == 1
                inp = inp[0]
            self.inputs[name] = inp
            self.outputs[name] = out

        return hook

    def clear(self):
        self.inputs.clear()
        self.outputs.clear()

    def take(self):
        """Hand over the tensors of the latest forward and start a new generation (the hooks look the dicts up on
        every call, so the next forward fills the fresh ones)."""
        got = (self.inputs, self.outputs)
        self.inputs, self.outputs = TapDict(), TapDict()
        return got

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []


def cosine_lrs(base_lr: float, t_max: int, n: int) -> List[float]:
    """Learning rate used by update 0..n-1 under ``CosineAnnealingLR(T_max=t_max)`` stepped once
    per update, evaluated with torch's recursive form so the doubles match the reference
    (:358, :375)."""
    lrs, lr = [], base_lr
    for t in range(n):
        if t > 0:
            if (t - 1 - t_max) % (2 * t_max) == 0:
                lr = lr + base_lr * (1 - math.cos(math.pi / t_max)) / 2
            else:
                lr = (1 + math.cos(math.pi * t / t_max)) / (1 + math.cos(math.pi * (t - 1) / t_max)) * lr
        lrs.append(lr)
    return lrs


# ------------------------------------------------------------------------------------------ per-layer plan
class _LayerPlan:
    __slots__ = ("name", "mod", "is_conv", "w", "b", "gw", "gb", "gw2", "gb2", "in_maps", "out_maps", "halves", "w_shape",
                 "off_w", "kpos", "vendor")


def _identity_block(n: int):
    e = torch.empty(0, dtype=torch.long)
    return (torch.arange(n), torch.arange(n), e, e)


def _fallback_out_width(separate_classifier: bool, model_type: str, num_classes: int) -> int:
    if not separate_classifier:
        return num_classes
    widths = {"rn50": 2048, "rn101": 2048, "rn20": 1024, "rn18": 512}
    if model_type not in widths:
        raise ValueError("Unknown model type: %s" % model_type)
    return widths[model_type]


MERGING_MODES = ("perm_gradmask", "reg_mean", "perm_separatels", "perm_mixedls")


def merging_mode(merging: str) -> str:
    """The reference tests ``merging`` with ``==`` for reg_mean and with ``in`` for the two stacked modes
    (pleas_merging.py:125, :132, :139); everything else is the default channel-merged form (:146-147)."""
    if merging == "reg_mean":
        return "reg_mean"
    if "perm_separatels" in merging:
        return "perm_separatels"
    if "perm_mixedls" in merging:
        return "perm_mixedls"
    return "perm_gradmask"


def half_maps(mode: str, bi, bo, cin: int, cout_src: int, device):
    """Per HALF-BATCH ``(input maps, output maps)`` -- each ``(row1, row2, n_merged)`` as ``block_maps`` gives them -- of
    the layer's regression problem (reference :125-147).  ``perm_gradmask`` has one half: the channel-merged input and
    target.  The other modes stack TWO half-batches along the sample axis, which here are two entries of the grouped
    launches that share the layer's weights (their gradients are summed):
      reg_mean         [ip1 ; ip2] -> [o1 ; o2], no permutation (source widths)
      perm_separatels  [i11, i1c, 0 ; i22, 0, i2c]                 -> [o11, o1c, 0 ; o22, 0, o2c]
      perm_mixedls     [(i11+i22)/2, i1c, 0 ; (i11+i22)/2, 0, i2c] -> the same targets
    Absent blocks are -1 rows: the kernels write / gather zeros there."""
    i32 = lambda t: t.to(device=device, dtype=torch.int32)
    neg = lambda n: torch.full((int(n),), -1, dtype=torch.int32, device=device)
    if mode == "perm_gradmask":
        return [(block_maps(bi, device), block_maps(bo, device))]
    if mode == "reg_mean":
        ai, ao = torch.arange(cin, dtype=torch.int32, device=device), torch.arange(cout_src, dtype=torch.int32, device=device)
        return [((ai, neg(cin), 0), (ao, neg(cout_src), 0)), ((neg(cin), ai, 0), (neg(cout_src), ao, 0))]

    def side(b, which, averaged):
        b1, b2, b1c, b2c = (i32(t) for t in b)
        n = b1.numel() + b1c.numel() + b2c.numel()
        if which == 0:      # [x11 (or the average), x1c, 0]
            r1 = torch.cat([b1, b1c, neg(b2c.numel())])
            r2 = torch.cat([b2, neg(b1c.numel() + b2c.numel())]) if averaged else neg(n)
        else:               # [x22 (or the average), 0, x2c]
            r2 = torch.cat([b2, neg(b1c.numel()), b2c])
            r1 = torch.cat([b1, neg(b1c.numel() + b2c.numel())]) if averaged else neg(n)
        return r1.contiguous(), r2.contiguous(), (int(b1.numel()) if averaged else 0)

    mixed = mode == "perm_mixedls"
    return [(side(bi, h, mixed), side(bo, h, False)) for h in (0, 1)]


def dp_slice(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank's share of one batch under data-parallel PLeaS: contiguous, equal sample chunks."""
    if world == 1:
        return x
    if x.shape[0] % world:
        raise RuntimeError("batch of %d samples does not split over %d ranks" % (x.shape[0], world))
    return x.chunk(world)[rank]


def dp_sum_(flat: torch.Tensor, world: int, share: float = 1.0) -> torch.Tensor:
    """Sum of the flat gradient arena over ranks (ONE collective per update).  Each rank scales its
    residual by 2 / (local numel * world), so the sum IS the gradient of the full-batch mean."""
    from .activation_matching import _force_collectives

    if world > 1 or _force_collectives():
        import torch.distributed as dist

        if dist.is_initialized():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        else:                           # PLEAS_EMULATE_WORLD (activation_matching._dist_info): no peers
            _emulated_collective(flat, share)
    return flat


def dp_reduce_scatter_(flat: torch.Tensor, mine: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Sum of ``flat`` over ranks, of which this rank needs only its ``rank``-th 1/world slice: ONE reduce-scatter into
    ``mine`` (RCCL); backends without it (gloo, the single-GPU rehearsals) all-reduce and take the slice."""
    import torch.distributed as dist

    n = flat.numel() // world
    if not dist.is_initialized():              # PLEAS_EMULATE_WORLD: no peers
        return flat[rank * n:(rank + 1) * n]
    if dist.get_backend() == "nccl":
        dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM)
        return mine
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat[rank * n:(rank + 1) * n]


def dp_all_gather_(flat: torch.Tensor, rank: int, world: int) -> None:
    """Every rank's updated 1/world slice of ``flat`` to all ranks, in place (ONE all-gather)."""
    import torch.distributed as dist

    if not dist.is_initialized():
        return
    n = flat.numel() // world
    mine = flat[rank * n:(rank + 1) * n].clone()       # the collective must not read from its own output
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(flat, mine)
    else:
        dist.all_gather(list(flat.chunk(world)), mine)


_SPIN = {}


def _emulated_collective(flat: torch.Tensor, share: float = 1.0) -> None:
    """Profiling aid: with PLEAS_EMULATE_ALLREDUCE_US=T the stream is held for ``share`` * T microseconds where the
    all-reduce would run (a spin kernel on one CU; T = the whole gradient arena), so that what overlaps a collective
    can be studied on one GPU."""
    import os

    us = float(os.environ.get("PLEAS_EMULATE_ALLREDUCE_US", "0") or 0) * share
    if us <= 0 or not flat.is_cuda:
        return
    if "per_us" not in _SPIN:           # calibrate the spin kernel's cycle unit once
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        a.record()
        torch.cuda._sleep(2_000_000)
        b.record()
        b.synchronize()
        _SPIN["per_us"] = 2_000_000 / (a.elapsed_time(b) * 1e3)
    torch.cuda._sleep(int(us * _SPIN["per_us"]))


class FrozenSources:
    """The two frozen source models as the PLeaS loop runs them: eval mode, BatchNorm / add / ReLU chains folded into one
    HIP pass each (``source_forward.py``), taps on every Conv2d / Linear, each model on its own side stream.  A forward
    produces a GENERATION ``(device batch, taps of model1, taps of model2, events)``; ``queue`` holds the generations whose
    forwards are enqueued, in update order, as ``(batch as passed,) + generation``.

    Nothing here depends on the permutation or on the merged model, so the object can be built -- and the first forwards
    enqueued (``prefetch``) -- while the LAP kernel is still running; ``PleasFitter(sources=...)`` then takes it over."""

    def __init__(self, model1: nn.Module, model2: nn.Module, data_parallel: bool = False, fuse: bool = True,
                 overlap: bool = True, fused=None):
        from .. import hip_ops
        from .activation_matching import _dist_info

        self.ops = hip_ops
        self.rank, self.world = _dist_info() if data_parallel else (0, 1)
        self.device = next(iter(model1.parameters())).device
        if self.device.type != "cuda":
            raise hip_ops.PleasHipError("pleas_merging.train: source models must be on the GPU (got %s)" % self.device)
        self.tap1, self.tap2 = ActivationTap(model1), ActivationTap(model2)
        model1.eval()
        model2.eval()
        # the hooked Conv2d / Linear modules are shared with the rewritten graphs, so the taps see the same tensors
        self.src1, self.src2 = model1, model2
        if fused is not None:               # built ahead of time (prepare_sources)
            self.src1, self.src2 = fused
        elif fuse:
            self.src1, self.src2 = prepare_sources(model1, model2)
        # The two source forwards are independent chains of small kernels (one conv of a batch-16 ResNet fills a
        # fraction of 256 CUs).  Each source gets its own stream: their chains then also run NEXT TO the big grouped
        # kernels of an update when they are enqueued ahead of it.  Equal priorities: favouring either side was slower
        # (chain first 8.7 s, update first 11.5 s).
        self._side_streams = (hip_ops.role_stream(self.device, "model1"),
                              hip_ops.role_stream(self.device, "model2")) if overlap else None
        self._src_events = None
        self._slice_batch = True   # data parallel: every update splits its batch's samples over the ranks
        self.queue: collections.deque = collections.deque()
        self._staged: dict = {}    # id(pinned host batch) -> (batch, device copy, event): copies started ahead of their forward

    def stage(self, batches) -> None:
        """Start the host -> device copies of PINNED host batches now (copy stream); the forward that takes a batch later
        only waits for its event.  ``steps`` stages the group AFTER the one it launches: enqueued together with a group's
        forwards, a copy would sit behind that group's kernels in a shared hardware queue and surface when it is needed."""
        for b in batches:
            if torch.is_tensor(b) and not b.is_cuda and b.is_pinned() and id(b) not in self._staged:
                self._staged[id(b)] = (b,) + self.ops.h2d_start(b, self.device)

    def _to_device(self, b: torch.Tensor) -> torch.Tensor:
        hit = self._staged.pop(id(b), None) if torch.is_tensor(b) else None
        if hit is not None and hit[0] is b:
            return self.ops.h2d_finish(hit[1], hit[2], self.device)
        return self.ops.to_device_async(b, self.device)

    def launch(self, x: torch.Tensor, parts: int = 1, after_current: bool = True):
        """Enqueue both source forwards for ``x``; returns the generation (device batch, taps, events) the update will
        consume.  ``parts`` > 1: ``x`` is that many batches back to back, each of which is sample-sliced on its own."""
        x = self._to_device(x)
        if self._slice_batch:
            if parts > 1 and self.world > 1:
                h = x.shape[0] // parts
                x = torch.cat([dp_slice(x[i * h:(i + 1) * h], self.rank, self.world) for i in range(parts)], 0)
            else:
                x = dp_slice(x, self.rank, self.world)
        self.run_eager(x, after_current)
        events = self._src_events
        self._src_events = None
        return (x, self.tap1.take(), self.tap2.take(), events)

    @torch.no_grad()
    def launch_group(self, run: List[torch.Tensor], after_current: bool = True) -> None:
        """One source forward for several batches; their generations (views of its taps) join the queue."""
        first = self._side_streams[0] if (self._side_streams is not None and not after_current) else None
        with (torch.cuda.stream(first) if first is not None else contextlib.nullcontext()):
            # pinned host batches travel on a copy stream while the previous group's updates run (hip_ops.to_device_async)
            both = torch.cat([self._to_device(b) for b in run], 0)
            xdev, (in1, out1), (in2, out2), events = self.launch(both, parts=len(run), after_current=after_current)
        n = xdev.shape[0] // len(run)
        for i, b in enumerate(run):
            lo, hi = i * n, (i + 1) * n
            self.queue.append((b, xdev, (_TapView(in1, lo, hi), _TapView(out1, lo, hi)),
                               (_TapView(in2, lo, hi), _TapView(out2, lo, hi)), events))

    @torch.no_grad()
    def prefetch(self, batches, group: Optional[int] = None, max_groups: int = 1, memory_fraction: float = 0.5) -> int:
        """Enqueue the source forwards of the first batches NOW, on the side streams only -- not ordered after the work
        already queued on the current stream, which is the point: called from ``activation_matching(while_solving=...)``
        they run beside the LAP kernel (one workgroup per problem, 71 of 256 CUs, 0.28 s).  ``batches`` must be tensors
        whose values are already in place (resident inputs, or host tensors); ``PleasFitter.steps`` must later be given
        the same tensor objects first, in the same order.  Takes whole groups of ``group`` (default ``2 * world``) equal
        batches, at most ``max_groups`` of them and as long as their taps (sized from the first group) fit into
        ``memory_fraction`` of the HBM that is free or idle in the allocator's pools.  Returns the number of batches taken."""
        if self._side_streams is None:
            return 0
        group = max(1, int(group or 2 * self.world))
        batches = list(batches)
        # The side streams skip the current stream's queue (the LAP kernel), but the memory they are about to take from
        # their allocator pools may have been released by the host while its last readers are still queued: updates of an
        # earlier fitter on the update stream (waited for here), matching kernels on model1's stream (in order with it;
        # model2's stream follows model1's in run_eager).
        self._side_streams[0].wait_stream(self.ops.role_stream(self.device, "updates"))
        # what the taps may take: a share of the HBM that is free or sits unused in the caching allocator's pools
        idle = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        budget = (torch.cuda.mem_get_info(self.device)[0] + idle) * memory_fraction
        per_group, taken = None, 0
        for g in range(max_groups):
            run = batches[g * group:(g + 1) * group]
            if len(run) < group or any(not torch.is_tensor(b) or b.shape != run[0].shape or b.shape[0] == 0 for b in run):
                break
            if per_group is not None and (g + 1) * per_group > budget:
                break
            if group > 1:
                self.launch_group(run, after_current=False)
            else:
                self.queue.append((run[0],) + self.launch(run[0], after_current=False))
            taken += group
            if per_group is None:      # bytes one group's taps hold (inputs and outputs of every hooked layer, both models)
                gen = self.queue[-1]
                per_group = 0
                for taps in (gen[2][0], gen[2][1], gen[3][0], gen[3][1]):
                    base = taps.base if isinstance(taps, _TapView) else taps
                    per_group += sum(t.numel() * t.element_size() for t in base.values())
        return taken

    @torch.no_grad()
    def run_eager(self, x: torch.Tensor, after_current: bool = True) -> None:
        """Both source forwards (frozen: never under autograd), dispatched op by op; the hooks fill ``tap1`` / ``tap2`` (callers that read them right
        away must synchronise: the models run on side streams).  ``after_current=False``: the side streams are not
        ordered after the current stream's queue (``prefetch``); model2's stream then follows model1's up to here, which
        is where the batch was put together."""
        side = self._side_streams
        if side is None:
            with self.ops.pin_stream():
                self.src1(x)
                self.src2(x)
            return
        main = torch.cuda.current_stream(self.device)
        # The sources' tensors live in the side streams' allocator pools.  They are consumed on `main` (by the update
        # that owns this generation) and released by the host once that update is enqueued; a side stream re-uses them
        # only after a LATER wait_stream(main), i.e. after every consumer enqueued on `main` up to then -- with one
        # batch of look-ahead that is the update two generations back.  No record_stream bookkeeping needed.
        events = []
        if not after_current:          # model2's stream follows model1's up to HERE (the batch is in place), not its forward
            here = torch.cuda.Event()
            here.record(side[0])
            side[1].wait_event(here)
        for stream, model in zip(side, (self.src1, self.src2)):
            if after_current:
                stream.wait_stream(main)
            with torch.cuda.stream(stream), self.ops.pin_stream():
                model(x)
            ev = torch.cuda.Event()
            ev.record(stream)
            events.append(ev)
        self._src_events = events

    def close(self) -> None:
        self.tap1.remove()
        self.tap2.remove()
        self.queue.clear()
        self._staged.clear()


class PleasFitter:
    """State of one PLeaS run: flat arenas + per-layer plans.  ``train`` drives it; the bench
    uses it directly to time single steps."""

    def __init__(self, model1, model2, model3, spec, perm, costs, budget_ratios, max_steps: int, lr: float = 5e-4,
                 separate_classifier=False, num_classes=1000, model_type="rn50", data_parallel: bool = False,
                 fuse_sources: bool = True, overlap_sources: bool = True, fused_sources=None,
                 sources: Optional[FrozenSources] = None, merging: str = "perm_gradmask",
                 shard_optimizer: Optional[bool] = None):
        from .. import hip_ops

        self.ops = hip_ops
        # data parallel: every update splits the batch's samples over the ranks, gradients (one flat
        # arena) are summed with ONE all-reduce, so all ranks apply the same full-batch update
        self.sources = sources if sources is not None else FrozenSources(
            model1, model2, data_parallel=data_parallel, fuse=fuse_sources, overlap=overlap_sources, fused=fused_sources)
        self.rank, self.world, self.device = self.sources.rank, self.sources.world, self.sources.device
        self.model1, self.model2, self.model3 = model1, model2, model3
        self.merging = merging_mode(merging)
        self.stacked = self.merging != "perm_gradmask"       # two half-batches per layer (reference :125-144)
        blocks = get_blocks(spec, perm, costs, budget_ratios, False)
        self.perm_blocks = spread_blocks(spec, blocks)
        # The update's own launches go to a dedicated stream, not to the caller's: measured on the 401-update job,
        # work on torch's default (null) stream overlaps the side streams markedly worse than work on a created stream
        # (8.24 s vs 7.73 s for the whole job).  step() orders that stream after the caller's and the caller's after it.
        self._upd_stream = hip_ops.role_stream(self.device, "updates") if self.sources._side_streams is not None else None
        self._cur_x = None
        self.t1_in = self.t1_out = self.t2_in = self.t2_out = None   # taps of the update being applied
        # Per input shape: the merged inputs and residuals of every layer live in buffers that are kept between updates
        # (written and read on the update stream only), so the item tables of the three grouped launches change between
        # two updates of that shape in the tap addresses alone -- see _step.
        self._buffers: Dict[tuple, Dict[tuple, Tuple[torch.Tensor, torch.Tensor]]] = {}
        self._bufs: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}      # (layer, half-batch) -> (merged input, residual)
        self._replay = None      # (shape key, names, merge table, forward table, bias work) of the latest full update
        self.fast_updates = 0    # updates applied by patching the tables (the rest took the layer-by-layer path)

        layers = {n: m for n, m in model3.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
        # The grouped HIP kernels take dense, undilated Conv2d layers with a square kernel / stride / padding (and Linear
        # layers on 2-D inputs) -- every layer of the ResNets the reference loads.  Any other geometry (rectangular, dilated or
        # grouped Conv2d; Linear on inputs with more than two axes) is fitted the way the reference fits EVERY layer
        # (`layer(ip)` under autograd, pleas_merging.py:281-287) on the vendor's operators, layer by layer (_fit_layer_vendor);
        # an update that contains such a layer takes the layer-by-layer path every time ("Layer geometries").
        self.vendor_layers = [n for n, m in layers.items() if isinstance(m, nn.Conv2d) and not _square_conv(m)]
        self.layer_modules = layers
        pad4 = lambda n: (n + 3) // 4 * 4   # every tensor starts 16-byte aligned inside the arenas (vector loads)
        total = sum(pad4(p.numel()) for m in layers.values() for p in m.parameters())
        # shard_optimizer (data parallel; default ON there since round 4, DESIGN.md section 5: the per-update exchange is as
        # long as a rank's per-update compute, so nothing that does not shrink with the ranks may stay on the update's
        # path): rank r keeps the Adam moments of -- and applies the update to -- the r-th
        # 1/world slice of the flat arena only: reduce-scatter of the gradients, Adam on the slice, all-gather of the
        # parameters.  Same bytes on the links as the all-reduce, 1/world of the optimiser's work and state per rank.
        from .activation_matching import _force_collectives

        if shard_optimizer is None:
            shard_optimizer = True
        self.shard_optimizer = bool(shard_optimizer) and (self.world > 1 or _force_collectives())
        if self.shard_optimizer:
            total = (total + 4 * self.world - 1) // (4 * self.world) * (4 * self.world)    # equal, 16-byte aligned slices
        self._shard = total // self.world if self.shard_optimizer else total
        dev = self.device
        self.p = torch.zeros(total, dtype=torch.float32, device=dev)
        # the gradient arena carries this update's per-layer losses in its tail: ONE collective per update sums both
        self._g_ext = torch.zeros(total + len(layers), dtype=torch.float32, device=dev)
        self.g = self._g_ext[:total]
        # stacked modes: the second half-batch's weight gradients land in an arena of their own (two entries of one grouped
        # launch must not write the same tensor) and are added before the exchange / the optimiser
        self.g2 = torch.zeros(total, dtype=torch.float32, device=dev) if self.stacked else None
        self.m = torch.zeros(self._shard, dtype=torch.float32, device=dev)     # this rank's slice only when sharded
        self.v = torch.zeros(self._shard, dtype=torch.float32, device=dev)
        self._g_shard = torch.zeros(self._shard, dtype=torch.float32, device=dev) if self.shard_optimizer else None
        self.mask = torch.ones(total, dtype=torch.float32, device=dev)   # frozen blocks are zeroed in place below
        self.plans: List[_LayerPlan] = []
        off, k = 0, 0
        for name, mod in layers.items():
            plan = _LayerPlan()
            plan.name, plan.mod, plan.is_conv = name, mod, isinstance(mod, nn.Conv2d)
            plan.vendor = name in self.vendor_layers
            plan.b = plan.gb = plan.gw2 = plan.gb2 = None
            # k x k convolutions with Cin % 32 == 0 keep their weight (gradient, mask, Adam state) KERNEL-POSITION-MAJOR
            # [Cout][KH][KW][Cin] in the arenas: the fused forward then has one tap per K chunk, the weight-gradient
            # kernel writes that layout directly, and the elementwise optimiser does not care (finish() permutes back)
            plan.kpos = bool(plan.is_conv and _square_conv(mod) and mod.kernel_size[0] > 1
                             and mod.weight.shape[1] % 32 == 0)   # the tensor's width: module attributes may be stale
            for pname, prm in mod.named_parameters():
                n = prm.numel()
                kp = plan.kpos and pname == "weight"
                shape = (prm.shape[0], prm.shape[2], prm.shape[3], prm.shape[1]) if kp else tuple(prm.shape)
                # model3 may live on the host: move first, permute on the device (a strided copy of a 3x3 weight on the CPU
                # costs milliseconds), and write the mask's frozen blocks straight into the arena
                src = prm.detach().to(dev, non_blocking=True)
                view, gview = self.p[off:off + n].view(shape), self.g[off:off + n].view(shape)
                view.copy_(src.permute(0, 2, 3, 1) if kp else src)
                if prm.dim() >= 2:
                    mview = self.mask[off:off + n].view(shape)
                    for rows, cols in _frozen_blocks(self.perm_blocks, name):
                        if kp:
                            mview[rows, :, :, cols] = 0.0
                        else:
                            mview[rows, cols] = 0.0
                g2view = self.g2[off:off + n].view(shape) if self.stacked else None
                if pname == "weight":
                    plan.w, plan.gw, plan.w_shape, plan.off_w = view, gview, tuple(prm.shape), off
                    plan.gw2 = g2view
                else:
                    plan.b, plan.gb = view, gview
                    plan.gb2 = g2view
                off += pad4(n)
                k += 1
            src = get_attr(model1, name.split("."))
            cin = src.in_channels if plan.is_conv else src.in_features
            bi = self.perm_blocks.get(Axis("%s.weight" % name, 1)) or _identity_block(cin)
            bo = self.perm_blocks.get(Axis("%s.weight" % name, 0)) or _identity_block(
                _fallback_out_width(separate_classifier, model_type, num_classes))
            cout_src = src.out_channels if plan.is_conv else src.out_features
            plan.halves = half_maps(self.merging, bi, bo, cin, cout_src, dev)
            plan.in_maps, plan.out_maps = plan.halves[0]
            self.plans.append(plan)
        self.max_steps = max_steps
        self.lrs = cosine_lrs(lr, max_steps, max_steps + 1)
        self.step_count = 0
        self.loss_now = self._g_ext[total:]                                             # this step, per layer
        self.loss_sum = torch.zeros(len(self.plans), dtype=torch.float32, device=dev)   # since last report
        self.wgrad = hip_ops.WgradBatch(dev)
        self.fwd = hip_ops.FwdBatch(dev)
        self.merge = hip_ops.MergeBatch(dev)
        self._fwd_loss = None
        self._fwd_index = None

    # the source side lives in `self.sources`; these names are what the loop (and the tests) use
    tap1 = property(lambda self: self.sources.tap1)
    tap2 = property(lambda self: self.sources.tap2)
    src1 = property(lambda self: self.sources.src1)
    src2 = property(lambda self: self.sources.src2)
    _queue = property(lambda self: self.sources.queue)
    _side_streams = property(lambda self: self.sources._side_streams)

    @property
    def _slice_batch(self):
        return self.sources._slice_batch

    @_slice_batch.setter
    def _slice_batch(self, value):
        self.sources._slice_batch = bool(value)

    def _launch_sources(self, x: torch.Tensor, parts: int = 1):
        return self.sources.launch(x, parts)

    def _run_sources(self, x: torch.Tensor) -> None:
        self.sources.run_eager(x)

    # -- one layer: merged input; queue forward(+target+residual+loss) and weight gradient for the grouped launches
    def _fit_layer(self, idx: int, plan: _LayerPlan) -> None:
        ops = self.ops
        name = plan.name
        ip1, ip2 = self.t1_in[name], self.t2_in[name]
        o1, o2 = self.t1_out[name], self.t2_out[name]
        mod = plan.mod
        cout = plan.w_shape[0]
        square = plan.is_conv and _square_conv(mod)
        linear = (not plan.is_conv) and ip1.dim() == 2
        if not (square or linear):
            self._fit_layer_vendor(idx, plan)
            return
        geo = (tuple(mod.kernel_size), mod.stride[0], mod.padding[0]) if plan.is_conv else ((1, 1), 1, 0)
        # a 1x1 convolution with a stride reads every s-th pixel of every s-th line: the grouped merge writes exactly those, and
        # the layer is a dense 1x1 stride-1 layer for the forward and the weight gradient (16-byte loads along the pixel axis
        # instead of 4-byte loads at a stride; ResNet-101's three downsample layers: 52 -> ~100 TFLOP/s in the grouped forward)
        sub = geo[1] if (plan.is_conv and geo[0] == (1, 1) and geo[1] > 1 and geo[2] == 0) else 1
        if sub > 1:
            geo = ((1, 1), 1, 0)
        halves = len(plan.halves)
        for h, (in_maps, (r1, r2, nm)) in enumerate(plan.halves):
            if cout != r1.numel():
                raise RuntimeError("layer %s: %d merged outputs vs %d target blocks" % (name, cout, r1.numel()))
            if in_maps[0].numel() != plan.w_shape[1]:
                raise RuntimeError("layer %s: %d merged inputs vs %d input blocks" % (name, plan.w_shape[1], in_maps[0].numel()))
            # merged input: queued for the ONE grouped merge launch that precedes the grouped forward (merge.flush)
            kept = self._bufs.get((idx, h))
            ip = self.merge.add(ip1, ip2, 1, *in_maps, out=kept[0] if kept else None, subsample=sub)
            resid = kept[1] if kept else torch.empty((ip.shape[0], cout) + tuple(o1.shape[2:]), dtype=torch.float32,
                                                     device=ip.device)
            if kept is None:
                self._bufs[(idx, h)] = (ip, resid)
            n = resid.numel() * halves * self.world    # the mean runs over the full (global, stacked) batch
            self.fwd.add(ip, plan.w, plan.b, o1, o2, r1, r2, nm, resid, 2.0 / n, 1.0 / n, *geo,
                         flags=ops.FwdBatch.KPOS_MAJOR if plan.kpos else 0)
            self._fwd_rows.append(idx)
            gw, gb = (plan.gw, plan.gb) if h == 0 else (plan.gw2, plan.gb2)   # second half: its own arena, added in _step
            # every layer, the 3-channel stem included (its rows are (channel, tap) pairs: "virtual channels", csrc/conv.hip)
            self.wgrad.add(resid, ip, gw, *geo, flags=ops.WgradBatch.KPOS_MAJOR if plan.kpos else 0)
            if gb is not None:
                self._bias_grads.append((resid, plan, gb))

    def _fit_layer_vendor(self, idx: int, plan: _LayerPlan) -> None:
        """A layer the grouped kernels do not take, fitted as the reference fits every layer (pleas_merging.py:116-147, :281-287):
        merged input and target by ``pleas_merge_blocks``, then ``layer(ip)``, the MSE and its gradients on the vendor's operators
        under autograd.  Gradients land in the same arena (mask and Adam are shared with all other layers); the loss joins
        ``loss_now`` after the grouped forward's."""
        import torch.nn.functional as F

        ops, name, mod = self.ops, plan.name, plan.mod
        ip1, ip2 = self.t1_in[name], self.t2_in[name]
        o1, o2 = self.t1_out[name], self.t2_out[name]
        halves = len(plan.halves)
        for h, (in_maps, out_maps) in enumerate(plan.halves):
            axis = 1 if plan.is_conv else ip1.dim() - 1      # channels: axis 1 of a feature map, the last axis of a Linear input
            ip = ops.merge_blocks(ip1, ip2, axis, *in_maps)
            op = ops.merge_blocks(o1, o2, axis, *out_maps)
            with torch.enable_grad():
                w = plan.w.detach().requires_grad_(True)
                b = plan.b.detach().requires_grad_(True) if plan.b is not None else None
                if plan.is_conv:
                    out = F.conv2d(ip, w, b, mod.stride, mod.padding, mod.dilation, mod.groups)
                else:
                    out = F.linear(ip, w, b)
                n = out.numel() * halves * self.world      # the mean runs over the full (global, stacked) batch
                loss = ((out - op) ** 2).sum() / n
                grads = torch.autograd.grad(loss, [w] + ([b] if b is not None else []))
            gw, gb = (plan.gw, plan.gb) if h == 0 else (plan.gw2, plan.gb2)
            gw.copy_(grads[0])
            if b is not None:
                gb.copy_(grads[1])
            self._vendor_losses.append((idx, loss.detach()))
        self._complete = False      # no table to replay: the next update goes layer by layer again

    def _bias_gradients(self) -> None:
        for resid, plan, gb in self._bias_grads:      # HIP: one deterministic row-sum launch per biased layer
            self.ops.channel_sum(resid if resid.dim() > 1 else resid.reshape(1, -1), gb)

    def _begin_update(self, x: torch.Tensor, next_x: Optional[torch.Tensor]) -> None:
        """Taps of ``x`` become current (running its sources now unless they were prefetched); the sources of
        ``next_x`` are enqueued BEFORE this update's own kernels, so that they overlap them on the side streams."""
        if self._queue and self._queue[0][0] is not x:
            raise RuntimeError("PleasFitter.step: a different batch was prefetched than the one passed now")
        cur = self._queue.popleft() if self._queue else (x,) + self._launch_sources(x)
        if next_x is not None and not self._queue and self._side_streams is not None:
            self._queue.append((next_x,) + self._launch_sources(next_x))
        _, self._cur_x, (self.t1_in, self.t1_out), (self.t2_in, self.t2_out), events = cur
        if events is not None:
            main = torch.cuda.current_stream(self.device)
            for ev in events:
                main.wait_event(ev)

    def _end_update(self) -> None:
        # the device copy of the batch was read on the side streams: it is released only now, after this update's
        # kernels (which waited for those streams) are enqueued on the stream that owns its memory
        self._cur_x = None
        self.t1_in = self.t1_out = self.t2_in = self.t2_out = None

    @contextlib.contextmanager
    def _session(self):
        """Make the fitter's own stream current; order it after the caller's stream on entry and the caller's after it
        on exit (so results are visible to whatever the caller enqueues next)."""
        upd = self._upd_stream
        if upd is None or torch.cuda.current_stream(self.device) == upd:
            yield
            return
        caller = torch.cuda.current_stream(self.device)
        upd.wait_stream(caller)
        try:
            with torch.cuda.stream(upd):
                yield
        finally:
            caller.wait_stream(upd)

    def steps(self, batches, lookahead: Optional[bool] = None, sources_per_forward: Optional[int] = None):
        """Run one update per tensor of ``batches``; yields the index of each finished update.  Between two updates the
        consumer's code runs on ITS OWN current stream, ordered after the update just applied (the fitter's stream is
        entered and left per update).

        Up to ``sources_per_forward`` consecutive batches of equal shape (1: every batch on its own) go through the
        frozen sources as ONE forward of the concatenated batch, and the updates read their slices (views) of its taps --
        the sources are in eval mode, so every sample's activations are what a separate forward gives (up to the vendor
        kernels' rounding at another batch size) and the update order is unchanged.  ResNet-101 pair, 16 samples per
        update: 6.6 ms of source forwards per update one by one, 6.2 ms in pairs, and the host dispatches them once per
        group (whole job 7.22 -> 6.94 s).  Groups of four bring nothing more (7.05 s) and every new batch size costs the
        vendor library its first-use set-up, so the default is two per rank: ``2 * world`` under data parallelism, where an
        update's share is ``batch / world`` samples -- the source forward then has the same size for every ``world``, and its
        ~5 ms of host dispatch (which bounds a rank at 6.8 ms per update when every update forwards its own 2 samples,
        ``tools/probe_dp_rank.py``) is paid once per group.  What is left when ``batches`` runs out forms one smaller group
        (a single left-over batch goes alone), so a job meets at most three forward shapes.

        ``lookahead=True``: the source forwards of the NEXT group (or batch) are enqueued before the current group's updates
        and run beside them on the side streams (two tap generations in flight).  On one GPU the update kernels already
        fill the chip: -3 % wall-clock (7.96 s vs 8.19 s, ungrouped) while every grouped kernel takes longer because it
        shares the CUs (fused forward 2.9 -> 4.3 ms per launch), so it is off there and per-kernel timings stay
        interpretable.  Under data parallelism (the default then) the stream that applies the updates idles during every
        gradient all-reduce, and the prefetched source forwards are what fills those gaps."""
        if sources_per_forward is None:
            sources_per_forward = 2 * self.world
        group = max(1, int(sources_per_forward))
        if lookahead is None:
            lookahead = self.world > 1
        keep = group if lookahead else 0      # sources are launched whenever no more than `keep` generations are queued
        it = iter(batches)
        for gen in list(self._queue):         # forwards enqueued beforehand (FrozenSources.prefetch): same batches first
            if next(it, None) is not gen[0]:
                raise RuntimeError("PleasFitter.steps: the prefetched batches must be the first ones, in order")
        ahead: List[torch.Tensor] = []        # fetched from `it`, sources not launched yet
        exhausted = False
        idx = 0
        while True:
            # The fitter's stream is current while an update is put together and ONLY then: the session is entered and left
            # per update, so a consumer that breaks out of (or raises inside) the loop is on its own stream, ordered after
            # the updates it has seen -- nothing is left to a generator's finalisation.
            with self._session():
                while len(self._queue) <= keep:
                    while len(ahead) < 2 * group and not exhausted:      # one group to launch, one whose copies start now
                        nxt = next(it, None)
                        if nxt is None:
                            exhausted = True
                        else:
                            ahead.append(nxt)
                    if not ahead:
                        break
                    run = [ahead[0]]
                    for cand in ahead[1:group]:
                        if torch.is_tensor(cand) and torch.is_tensor(run[0]) and cand.shape == run[0].shape and cand.shape[0] > 0:
                            run.append(cand)
                        else:
                            break
                    if group > 1 and (len(run) == group or (exhausted and len(run) > 1 and len(run) == len(ahead))):
                        # a full group -- or, when the batches have run out, what is left of one (one more forward size,
                        # instead of one forward per left-over batch)
                        self.sources.launch_group(run)      # one generation per batch of the run
                        del ahead[:len(run)]
                    else:
                        b = ahead.pop(0)
                        self._queue.append((b,) + self._launch_sources(b))
                    self.sources.stage(ahead[:group])
                if not self._queue:
                    break
                self.step(self._queue[0][0])
            yield idx
            idx += 1

    @torch.no_grad()
    def step(self, x: torch.Tensor, next_x: Optional[torch.Tensor] = None) -> None:
        """One update: reference ``step`` (:234-302) + ``lr_sched.step()`` (:375).  ``next_x`` (optional): the batch
        of the NEXT call; its source forwards are enqueued now and run beside this update.  On return the work is
        ordered before anything enqueued later on the caller's stream."""
        with self._session():
            self._step(x, next_x)

    def _update_key(self) -> tuple:
        first = self.plans[0].name
        if first not in self.t1_in:
            return ()
        taps = self.t1_in
        n = (taps.hi - taps.lo) if isinstance(taps, _TapView) else taps[first].shape[0]
        return (n,) + tuple(self._cur_x.shape[1:])

    def _step(self, x: torch.Tensor, next_x: Optional[torch.Tensor]) -> None:
        """One update.  The first update of an input shape goes layer by layer (_fit_layer: checks, buffers, item tables of
        the three grouped launches).  Every further update of that shape differs from it in the addresses of the source
        taps alone, so it rewrites those four pointer columns (numpy, from one address table per source forward) and
        launches again: ~2.3 ms -> ~0.5 ms of host time per update, which is what bounds a rank once its share of an update
        is a few samples."""
        self._begin_update(x, next_x)
        key = self._update_key()
        replay = self._replay if (self._replay is not None and key and self._replay[0] == key) else None
        ptrs = None
        if replay is not None:
            names = replay[1]
            ptrs = [tap_pointers(t, names) for t in (self.t1_in, self.t2_in, self.t1_out, self.t2_out)]
            if any(p is None for p in ptrs):
                replay = None
        if replay is not None:
            _, names, merge_tab, fwd_tab, self._bias_grads = replay
            merge_tab["w1"][:], merge_tab["w2"][:], fwd_tab["o1"][:], fwd_tab["o2"][:] = ptrs
            with self.ops.pin_stream():
                self.merge.relaunch()
                self.fwd.relaunch(self._fwd_loss)
            self.fast_updates += 1
        else:
            self._replay = None
            self._bufs = self._buffers.setdefault(key, {}) if key else {}
            self._fwd_rows, self._bias_grads = [], []
            self._vendor_losses, self._complete = [], True
            complete = bool(key)
            with self.ops.pin_stream():
                for idx, plan in enumerate(self.plans):
                    if plan.name not in self.t1_in or plan.name not in self.t2_in:
                        print("Key error on %s" % plan.name)
                        complete = False
                        continue
                    self._fit_layer(idx, plan)
            complete = complete and self._complete
            self.merge.flush()   # ONE grouped launch: the merged inputs of every layer
            if self._fwd_rows:   # ONE grouped MFMA launch: forward + target + residual + loss of every merged layer
                if self._fwd_loss is None or self._fwd_loss.numel() != len(self._fwd_rows):
                    self._fwd_loss = torch.zeros(len(self._fwd_rows), dtype=torch.float32, device=self.device)
                    self._fwd_index = torch.tensor(self._fwd_rows, dtype=torch.long, device=self.device)
                self.fwd.flush(self._fwd_loss)
        if self._fwd_rows:
            if self.stacked:        # two entries (half-batches) per layer: their losses add up
                self.loss_now.zero_()
                self.loss_now.index_add_(0, self._fwd_index, self._fwd_loss)
            else:
                self.loss_now.index_copy_(0, self._fwd_index, self._fwd_loss)
        if replay is None and self._vendor_losses:      # layers fitted on the vendor's operators (_fit_layer_vendor)
            if not self._fwd_rows:
                self.loss_now.zero_()
            seen = set()
            for idx, loss in self._vendor_losses:
                if idx in seen:
                    self.loss_now[idx] += loss
                else:
                    self.loss_now[idx] = loss
                    seen.add(idx)
        self._bias_gradients()
        if replay is not None:
            with self.ops.pin_stream():
                self.wgrad.relaunch()
        else:
            n_wgrad = len(self.wgrad._keep)
            self.wgrad.flush()  # ONE grouped MFMA launch: weight gradients of every merged layer
            if complete and n_wgrad > 0:
                names = tuple(self.plans[i].name for i in self._fwd_rows)
                merge_tab, fwd_tab = self.merge.table(), self.fwd.table()
                if merge_tab is not None and fwd_tab is not None and len(merge_tab) == len(fwd_tab) == len(names):
                    self._replay = (key, names, merge_tab, fwd_tab, self._bias_grads)
        if self.stacked:
            self.g.add_(self.g2)     # second half-batch's gradients (same launch, own arena)
        lr = self.lrs[min(self.step_count, len(self.lrs) - 1)]
        self.step_count += 1
        if self.shard_optimizer:
            lo, hi = self.rank * self._shard, (self.rank + 1) * self._shard
            g_mine = dp_reduce_scatter_(self.g, self._g_shard, self.rank, self.world)
            dp_sum_(self.loss_now, self.world)
            self.loss_sum.add_(self.loss_now)
            self.ops.masked_adam(self.p[lo:hi], g_mine, self.mask[lo:hi], self.m, self.v, lr, self.step_count)
            dp_all_gather_(self.p, self.rank, self.world)
        else:
            dp_sum_(self._g_ext, self.world)     # gradients + losses: ONE collective per update
            self.loss_sum.add_(self.loss_now)
            self.ops.masked_adam(self.p, self.g, self.mask, self.m, self.v, lr, self.step_count)
        self._end_update()

    def finish(self) -> nn.Module:
        """Write the fitted weights back into ``model3`` and drop the hooks (reference :392-403)."""
        sd = self.model3.state_dict()
        for plan in self.plans:
            sd["%s.weight" % plan.name] = (plan.w.detach().permute(0, 3, 1, 2) if plan.kpos else plan.w.detach()).clone()
            if plan.b is not None:
                sd["%s.bias" % plan.name] = plan.b.detach().clone()
        self.model3.load_state_dict(sd)
        self.sources.close()
        return self.model3


def train(dataloader, model1, model2, model3, spec, perm, costs, budget_ratios, WANDB, MAX_STEPS, wandb_run,
          separate_classifier=False, merging="perm_gradmask", num_classes=1000, lr=5e-4, verbose=False,
          model_type="rn50", solver="adam"):
    """Fit ``model3``'s Conv2d/Linear weights layer by layer (reference :305-405; same positional
    order and defaults; ``solver`` is the only addition).  Consumes one batch per update from a
    single pass over ``dataloader`` and performs ``MAX_STEPS + 1`` updates if it is long enough.
    Puts ``model1``/``model2`` in eval mode (as the reference), mutates and returns ``model3``.
    """
    if solver == "normal_eq":
        from .normal_eq import train_normal_eq

        return train_normal_eq(dataloader, model1, model2, model3, spec, perm, costs, budget_ratios, MAX_STEPS,
                               separate_classifier, num_classes, model_type, verbose, merging=merging)
    if solver != "adam":
        raise ValueError("solver must be 'adam' or 'normal_eq'")
    fit = PleasFitter(model1, model2, model3, spec, perm, costs, budget_ratios, MAX_STEPS, lr, separate_classifier,
                      num_classes, model_type, merging=merging)
    names = [p.name for p in fit.plans]

    def inputs():
        for idx, batch in enumerate(dataloader):
            if idx > MAX_STEPS:
                break
            yield batch[0]

    for idx in fit.steps(inputs()):
        if verbose:
            print(float(fit.loss_sum.sum()))
        if idx % 20 == 0 and idx:
            per_layer = (fit.loss_sum / 20).cpu()
            total = float(per_layer.sum())
            print("Loss: %.3f" % total)
            if WANDB:
                metrics = {"loss_%s" % n: float(v) for n, v in zip(names, per_layer)}
                metrics["step"], metrics["loss"] = idx, total
                wandb_run.log(metrics)
            fit.loss_sum.zero_()
    return fit.finish()
