"""Tensor-level wrappers over the C-ABI (include/pleas_hip.h).

PyTorch is used for device memory and streams only: every function here enqueues a
hand-written gfx950 kernel on torch's current stream via ctypes.  Inputs must be fp32
CUDA(=HIP) tensors; anything else raises -- there is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes
import math
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import EPI_INNER, EPI_NEG_CDIST, PleasHipError, check


class _Pinned(threading.local):
    """Per THREAD: torch's current stream is thread-local, so a handle pinned by one thread must never be seen by
    another (a DataLoader / collate thread, the autograd thread, a second fitter)."""

    depth = 0
    handle = 0


_PINNED = _Pinned()


def _stream() -> int:
    """Raw hipStream_t of torch's current stream.  ``torch.cuda.current_stream()`` costs ~8 us; inside a
    ``pin_stream()`` block (one PLeaS update / matching batch) the handle is looked up once -- by the thread that
    entered the block, for that thread only."""
    if _PINNED.depth > 0:
        return _PINNED.handle
    return torch.cuda.current_stream().cuda_stream


class pin_stream:
    """Context manager: all wrappers called by THIS thread inside reuse the stream handle that is current at entry
    (nested blocks re-pin and restore)."""

    def __enter__(self):
        self._saved = (_PINNED.depth, _PINNED.handle)
        _PINNED.depth += 1
        _PINNED.handle = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        _PINNED.depth, _PINNED.handle = self._saved
        return False


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if not t.is_cuda:
            raise PleasHipError("HIP path needs tensors on the GPU (got device %s); no CPU fallback" % t.device)
        if t.dtype != torch.float32:
            raise PleasHipError("HIP path computes in fp32 (got %s)" % t.dtype)


_ROLE_STREAMS = {}


def role_stream(device: torch.device, role: str, priority: int = 0) -> torch.cuda.Stream:
    """ONE HIP stream per (device, role) for the life of the process.  torch's caching allocator keeps freed blocks per
    allocation stream, so a job that created fresh streams could not reuse anything an earlier job (or a warm-up) had
    left behind: it went back to hipMalloc for its ~15 GB of taps per source forward, and the pools of the abandoned
    streams stayed reserved until an out-of-memory retry released them."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    st = _ROLE_STREAMS.get(key)
    if st is None:
        st = _ROLE_STREAMS[key] = torch.cuda.Stream(device, priority=priority)
    return st


def to_device_async(x: torch.Tensor, device: torch.device) -> torch.Tensor:
    """``x`` on ``device``, usable on the CURRENT stream.  The reference moves every batch inside its loops with a blocking
    ``x.cuda()`` (activation_matching.py:121, pleas_merging.py:266).  A PINNED host tensor is copied on a dedicated copy
    stream ("h2d") instead: the host enqueues a loop's kernels far ahead of the GPU, so the copy of batch b + 1 runs while
    batch b computes, and the current stream only waits for the copy's event.  Pageable host tensors and device tensors
    take the ordinary path."""
    device = torch.device(device)
    if x.device == device:
        return x
    if x.is_cuda or not x.is_pinned():
        return x.to(device, non_blocking=True)
    d, ev = h2d_start(x, device)
    return h2d_finish(d, ev, device)


def h2d_start(x: torch.Tensor, device: torch.device):
    """Enqueue the copy of a PINNED host tensor on the copy stream NOW; returns ``(device tensor, event)`` for
    :func:`h2d_finish`.  A caller that knows its next batches starts their copies a whole step early: a process's streams
    share four hardware queues, so a copy enqueued right before it is needed sits behind the kernels already queued there."""
    copy = role_stream(torch.device(device), "h2d")
    with torch.cuda.stream(copy):
        d = x.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(copy)
    return d, ev


def h2d_finish(d: torch.Tensor, ev, device: torch.device) -> torch.Tensor:
    """Make the CURRENT stream wait for a copy started by :func:`h2d_start` and hand the tensor over to it."""
    cur = torch.cuda.current_stream(torch.device(device))
    cur.wait_event(ev)
    d.record_stream(cur)      # allocated in the copy stream's pool, consumed on `cur`
    return d


class Workspace:
    """Grow-only device scratch buffer, one per (device, stream): two streams that run ``gram_accum`` / ``sqerr`` side
    by side must not share scratch.  Callers never see hidden allocations inside the C library."""

    _per_stream: dict = {}

    def __init__(self, device: torch.device):
        self.device = device
        self.buf = torch.empty(0, dtype=torch.uint8, device=device)

    @classmethod
    def get(cls, device: torch.device) -> "Workspace":
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), _stream())
        ws = cls._per_stream.get(key)
        if ws is None:
            ws = cls._per_stream[key] = cls(device)
        return ws

    def reserve(self, nbytes: int) -> torch.Tensor:
        if self.buf.numel() < nbytes:
            # the old buffer may still be in use by kernels queued on the current stream, which need not be the stream it
            # was allocated on: tell the caching allocator before dropping it
            if self.buf.numel():
                self.buf.record_stream(torch.cuda.current_stream(self.device))
            self.buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


def _as_bchw(t: torch.Tensor, axis: int) -> Tuple[int, int, int]:
    shape = t.shape
    axis = axis % t.dim()
    return math.prod(shape[:axis]), shape[axis], math.prod(shape[axis + 1:])


# ---------------------------------------------------------------------------------------- gram / cdist
def gram_ws_bytes(B: int, C: int, HW: int) -> int:
    return int(_lib.lib().pleas_gram_ws_bytes(B, C, HW))


def gram_accum(x: torch.Tensor, y: torch.Tensor, axis: int, acc: torch.Tensor, epilogue: int,
               accumulate: bool = True) -> torch.Tensor:
    """acc (+)= cross-features of ``x`` and ``y`` along ``axis`` (see pleas_gram_accum)."""
    _need_gpu(x, y, acc)
    if x.shape != y.shape:
        raise PleasHipError("cross features need equal shapes, got %s and %s" % (tuple(x.shape), tuple(y.shape)))
    x, y = x.contiguous(), y.contiguous()
    B, C, HW = _as_bchw(x, axis)
    if tuple(acc.shape) != (C, C) or not acc.is_contiguous():
        raise PleasHipError("acc must be a contiguous (%d, %d) tensor" % (C, C))
    need = gram_ws_bytes(B, C, HW)
    ws = Workspace.get(x.device).reserve(need)
    rc = _lib.lib().pleas_gram_accum(x.data_ptr(), y.data_ptr(), B, C, HW, epilogue, int(bool(accumulate)),
                                     acc.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "pleas_gram_accum")
    return acc


def cross_features_cdist(x: torch.Tensor, y: torch.Tensor, a: int) -> torch.Tensor:
    """HIP drop-in for the reference's ``cross_features_cdist`` plug point
    (pleas/methods/activation_matching.py:31-46): returns the C x C negative-distance matrix."""
    C = x.shape[a]
    out = torch.empty(C, C, dtype=torch.float32, device=x.device)
    return gram_accum(x, y, a, out, EPI_NEG_CDIST, accumulate=False)


def cross_features_inner_product(x: torch.Tensor, y: torch.Tensor, a: int) -> torch.Tensor:
    """HIP drop-in for ``cross_features_inner_product`` (activation_matching.py:14-28)."""
    C = x.shape[a]
    out = torch.empty(C, C, dtype=torch.float32, device=x.device)
    return gram_accum(x, y, a, out, EPI_INNER, accumulate=False)


class GramBatch:
    """Collects the tracked nodes of one forward pass and contracts them all in ONE grouped launch
    (``pleas_gram_batch``): ``add`` while the models run, ``flush`` once per batch.

    The operand tensors must stay unmodified until ``flush`` (the twin graph therefore runs ReLU
    out of place); they are released right after the launch is queued.
    """

    def __init__(self, group_mats: Sequence[torch.Tensor], epilogue: int):
        _need_gpu(*group_mats)
        self.mats = list(group_mats)
        self.epilogue = epilogue
        k = len(self.mats)
        self._acc = (ctypes.c_void_p * k)(*[m.data_ptr() for m in self.mats])
        self._gc = (ctypes.c_int * k)(*[m.shape[0] for m in self.mats])
        self._keep: list = []
        self._meta: list = []
        self._arr = None
        self._ws = None
        self.device = self.mats[0].device

    def add(self, x: torch.Tensor, y: torch.Tensor, axis: int, group: int) -> int:
        """Queue one contracted node; returns its index in this batch (the handle ``add_derived`` refers to)."""
        if x.shape != y.shape or not x.is_cuda or x.dtype != torch.float32 or y.dtype != torch.float32:
            raise PleasHipError("GramBatch.add: fp32 CUDA operands of equal shape expected")
        x, y = x.contiguous(), y.contiguous()
        self._keep.append((x, y))
        self._meta.append(_as_bchw(x, axis) + (group, None))
        return len(self._keep) - 1

    def add_derived(self, source: int, scale_x: torch.Tensor, shift_x: torch.Tensor, scale_y: torch.Tensor,
                    shift_y: torch.Tensor, group: int) -> int:
        """Queue a node whose operands are per-channel affine images ``scale * v + shift`` of node ``source``'s operands
        (an eval-mode BatchNorm of a tracked convolution): no contraction, the reduce pass derives its contribution."""
        if not 0 <= source < len(self._keep) or self._meta[source][4] is not None:
            raise PleasHipError("GramBatch.add_derived: source must be a contracted node of this batch")
        B, C, HW = self._meta[source][:3]
        for t in (scale_x, shift_x, scale_y, shift_y):
            if not t.is_cuda or t.dtype != torch.float32 or t.numel() != C or not t.is_contiguous():
                raise PleasHipError("GramBatch.add_derived: scale / shift must be contiguous fp32 CUDA vectors of length C")
        self._keep.append((scale_x, shift_x, scale_y, shift_y))
        self._meta.append((B, C, HW, group, source))
        return len(self._keep) - 1

    def flush(self, accumulate: bool = True, keep: bool = False) -> None:
        """Contract everything added since the last flush.  ``keep=True`` leaves the node list in place (a caller that
        contracts the same device tensors again)."""
        n = len(self._keep)
        if n == 0:
            return
        if self._arr is None or len(self._arr) != n:
            self._arr = (_lib.GramNode * n)()
        arr = self._arr
        for i, (t, (B, C, HW, g, src)) in enumerate(zip(self._keep, self._meta)):
            a = arr[i]
            a.B, a.C, a.HW, a.group = B, C, HW, g
            if src is None:
                a.x, a.y, a.derived, a.source = t[0].data_ptr(), t[1].data_ptr(), 0, 0
                a.scale_x = a.shift_x = a.scale_y = a.shift_y = None
            else:
                a.x = a.y = None
                a.derived, a.source = 1, src
                a.scale_x, a.shift_x, a.scale_y, a.shift_y = (v.data_ptr() for v in t)
        lib = _lib.lib()
        if self._ws is None:
            need = int(lib.pleas_gram_batch_ws_bytes(arr, n, self._gc, len(self.mats)))
            if need == 0:
                raise PleasHipError("pleas_gram_batch_ws_bytes rejected the node list: %s" % lib.pleas_last_error().decode())
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)  # dedicated: tables live in it
            self._fresh = 1
        rc = lib.pleas_gram_batch(arr, n, self._acc, self._gc, len(self.mats), self.epilogue, int(bool(accumulate)),
                                  self._ws.data_ptr(), self._ws.numel(), self._fresh, _stream())
        self._fresh = 0
        if rc == -12:  # node list changed shape: size the workspace again
            self._ws = None
            self.flush(accumulate, keep)
            return
        check(rc, "pleas_gram_batch")
        if not keep:
            self._keep.clear()
            self._meta.clear()

    def drop(self) -> None:
        self._keep.clear()
        self._meta.clear()


# ---------------------------------------------------------------------------------------- LAP
def solve_lsa_batched(costs: Sequence[torch.Tensor], maximize: bool = True) -> List[torch.Tensor]:
    """All assignment problems of a model pair in one launch; returns device int64 vectors."""
    if not costs:
        return []
    _need_gpu(*costs)
    mats = []
    for c in costs:
        if c.dim() != 2 or c.shape[0] != c.shape[1]:
            raise PleasHipError("square cost matrices expected, got %s" % (tuple(c.shape),))
        if c.shape[0] > _lib.LSAP_MAX_N:
            raise PleasHipError("n = %d exceeds PLEAS_LSAP_MAX_N = %d" % (c.shape[0], _lib.LSAP_MAX_N))
        mats.append(c.contiguous())
    outs = [torch.empty(m.shape[0], dtype=torch.int64, device=m.device) for m in mats]
    k = len(mats)
    cost_ptrs = (ctypes.c_void_p * k)(*[m.data_ptr() for m in mats])
    out_ptrs = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
    ns = (ctypes.c_int * k)(*[m.shape[0] for m in mats])
    rc = _lib.lib().pleas_lsap_batched(cost_ptrs, ns, k, int(bool(maximize)), out_ptrs, _stream())
    check(rc, "pleas_lsap_batched")
    return outs


def hip_solve_lsa(A: torch.Tensor, maximize: bool = True) -> torch.Tensor:
    """HIP drop-in for ``scipy_solve_lsa`` (pleas/core/solvers.py:18-33): CPU int64 ``col_ind``."""
    return solve_lsa_batched([A], maximize)[0].cpu()


def host_solve_lsa(A: torch.Tensor, maximize: bool = True) -> torch.Tensor:
    """``pleas_lsap_host``: the library's solver on a HOST cost matrix (fp32 / fp64 CPU tensor), synchronous; same
    ``col_ind`` as scipy.  For callers that keep their data on the host (weight matching of CPU state dicts); device
    tensors belong to ``hip_solve_lsa`` and are refused here -- nothing falls back from one to the other."""
    if A.is_cuda:
        raise PleasHipError("host_solve_lsa takes a CPU tensor; use hip_solve_lsa for device tensors")
    if A.dim() != 2 or A.shape[0] != A.shape[1] or A.shape[0] < 1:
        raise PleasHipError("square cost matrix expected, got %s" % (tuple(A.shape),))
    if A.dtype not in (torch.float32, torch.float64):
        A = A.double()
    A = A.contiguous()
    out = torch.empty(A.shape[0], dtype=torch.int64)
    check(_lib.lib().pleas_lsap_host(A.data_ptr(), int(A.dtype == torch.float64), A.shape[0], int(bool(maximize)),
                                     out.data_ptr()), "pleas_lsap_host")
    return out


# ---------------------------------------------------------------------------------------- bn + add + relu
def bn_act(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, res: Optional[torch.Tensor] = None,
           relu: bool = True) -> torch.Tensor:
    """``act(x * scale[c] + shift[c] (+ res))`` over dim 1 of a contiguous fp32 tensor: inference BatchNorm,
    residual add and ReLU of a frozen source model in one pass (``pleas_bn_act``)."""
    _need_gpu(x, scale, shift)
    if x.dim() < 2 or x.dtype != torch.float32:
        raise PleasHipError("bn_act needs an fp32 [N, C, ...] tensor")
    x = x.contiguous()
    if res is not None:
        if res.shape != x.shape or res.dtype != torch.float32:
            raise PleasHipError("bn_act: residual shape/dtype differs from x")
        res = res.contiguous()
    C = x.shape[1]
    # scale / shift [batches][C]: x holds that many batches back to back along dim 0, each with its own affine map
    batches = scale.shape[0] if scale.dim() == 2 else 1
    if scale.numel() != batches * C or shift.shape != scale.shape or not scale.is_contiguous() or not shift.is_contiguous():
        raise PleasHipError("bn_act: scale/shift must be contiguous with one entry per channel (and batch)")
    if x.shape[0] % batches:
        raise PleasHipError("bn_act: %d samples do not split into %d batches" % (x.shape[0], batches))
    y = torch.empty_like(x)
    ptr = res.data_ptr() if res is not None else None
    if batches == 1:
        rc = _lib.lib().pleas_bn_act(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), ptr, y.data_ptr(), x.shape[0], C,
                                     math.prod(x.shape[2:]), int(relu), _stream())
    else:      # the tracked pass without its optional outputs is this pass
        rc = _lib.lib().pleas_bn_act_tracked_batches(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), ptr, None, None,
                                                     y.data_ptr(), x.shape[0] // batches, batches, C,
                                                     math.prod(x.shape[2:]), int(relu), _stream())
    _lib.check(rc, "pleas_bn_act")
    return y


def bn_act_maxpool(x: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], kernel: Tuple[int, int],
                   stride: int, padding: int, relu: bool = True) -> torch.Tensor:
    """``max_pool2d(act(x * scale[c] + shift[c]), kernel, stride, padding)`` of a contiguous fp32 NCHW tensor in one pass
    (``pleas_bn_act_maxpool``); ``scale is None``: plain max pooling."""
    _need_gpu(x) if scale is None else _need_gpu(x, scale, shift)
    if x.dim() != 4 or x.dtype != torch.float32:
        raise PleasHipError("bn_act_maxpool needs an fp32 [N, C, H, W] tensor")
    x = x.contiguous()
    N, C, H, W = x.shape
    if scale is not None and (scale.numel() != C or shift.numel() != C or not scale.is_contiguous() or not shift.is_contiguous()):
        raise PleasHipError("bn_act_maxpool: scale/shift must be contiguous with one entry per channel")
    KH, KW = kernel
    if stride <= 0:
        raise PleasHipError("bn_act_maxpool: stride must be positive")
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    y = torch.empty((N, C, max(Ho, 0), max(Wo, 0)), dtype=torch.float32, device=x.device)
    rc = _lib.lib().pleas_bn_act_maxpool(x.data_ptr(), scale.data_ptr() if scale is not None else None,
                                         shift.data_ptr() if scale is not None else None, y.data_ptr(), N, C, H, W, KH, KW,
                                         stride, padding, int(relu), _stream())
    _lib.check(rc, "pleas_bn_act_maxpool")
    return y


def bn_act_tracked(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, res: Optional[torch.Tensor], relu: bool,
                   keep_bn: bool = True):
    """One pass, every node of the chain kept: returns ``(bn, sum, act)`` = ``(x * scale + shift, bn + res, relu(sum))``;
    ``sum`` is None without a residual and ``act`` is None without ``relu`` (``pleas_bn_act_tracked``).
    ``keep_bn=False``: the BatchNorm value itself is not needed by the caller (its matching cost is derived from the
    convolution node, ``GramBatch.add_derived``) and is not written unless it is the chain's last value."""
    _need_gpu(x, scale, shift)
    x = x.contiguous()
    if res is not None:
        if res.shape != x.shape or res.dtype != torch.float32:
            raise PleasHipError("bn_act_tracked: residual shape/dtype differs from x")
        res = res.contiguous()
    C = x.shape[1]
    # scale / shift [batches][C]: x holds that many batches back to back along dim 0, each with its own affine map
    batches = scale.shape[0] if scale.dim() == 2 else 1
    if scale.numel() != batches * C or shift.shape != scale.shape or not scale.is_contiguous() or not shift.is_contiguous():
        raise PleasHipError("bn_act_tracked: scale/shift must hold one entry per channel (and batch)")
    if x.shape[0] % batches:
        raise PleasHipError("bn_act_tracked: %d samples do not split into %d batches" % (x.shape[0], batches))
    # the last value of the chain goes to `y`; earlier ones to the optional outputs
    y_bn = torch.empty_like(x) if ((res is not None or relu) and keep_bn) else None
    y_sum = torch.empty_like(x) if (res is not None and relu) else None
    y = torch.empty_like(x)
    rc = _lib.lib().pleas_bn_act_tracked_batches(x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                 res.data_ptr() if res is not None else None,
                                                 y_bn.data_ptr() if y_bn is not None else None,
                                                 y_sum.data_ptr() if y_sum is not None else None, y.data_ptr(),
                                                 x.shape[0] // batches, batches, C, math.prod(x.shape[2:]), int(relu), _stream())
    check(rc, "pleas_bn_act_tracked_batches")
    if res is None:
        return (y_bn, None, y) if relu else (y, None, None)
    return (y_bn, y_sum, y) if relu else (y_bn, y, None)


def bn_train_fold(bn: "torch.nn.BatchNorm2d", x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Train-mode ``bn`` on ``x`` as a per-channel affine map: returns fp32 device vectors ``(scale, shift)`` with
    ``F.batch_norm(x, ..., training=True) == x * scale[c] + shift[c]`` and updates ``bn``'s running statistics and
    ``num_batches_tracked`` exactly as the module's forward would (``pleas_bn_train_fold``).  The caller applies the map
    with ``bn_act`` / ``bn_act_tracked``."""
    _need_gpu(x)
    if x.dim() < 2:
        raise PleasHipError("bn_train_fold needs an [N, C, ...] tensor")
    x = x.contiguous()
    n, C = x.shape[0], x.shape[1]
    inner = math.prod(x.shape[2:])
    if n * inner <= 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
    lib = _lib.lib()
    need = int(lib.pleas_bn_train_ws_bytes(n, C))
    ws = torch.empty(need // 8, dtype=torch.float64, device=x.device)
    out = torch.empty(2, C, dtype=torch.float32, device=x.device)
    track = bn.track_running_stats and bn.running_mean is not None
    ptr = lambda t: t.data_ptr() if t is not None else None
    for t in (bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != x.device):
            raise PleasHipError("bn_train_fold: BatchNorm parameters / buffers must be contiguous fp32 on x's device")
    rc = lib.pleas_bn_train_fold(x.data_ptr(), n, C, inner, ptr(bn.weight), ptr(bn.bias), float(bn.eps),
                                 -1.0 if bn.momentum is None else float(bn.momentum),
                                 ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                 ptr(bn.num_batches_tracked) if (track and bn.num_batches_tracked is not None) else None,
                                 out[0].data_ptr(), out[1].data_ptr(), ws.data_ptr(), need, _stream())
    check(rc, "pleas_bn_train_fold")
    return out[0], out[1]


class BnTrainFold:
    """``bn_train_fold`` for ONE BatchNorm2d called batch after batch (graph callable of the matching twin in train mode and
    of the BN-reset pass): workspace, output vectors and the module's parameter / buffer addresses are looked up once per
    input shape instead of per call -- those passes are bound by host dispatch (104 BatchNorm2d per ResNet-101 forward).
    The returned ``(scale, shift)`` are views of a buffer that the NEXT call overwrites: consume them on the same stream
    before folding the next batch (what a forward pass does)."""

    def __init__(self, bn: "torch.nn.BatchNorm2d"):
        self.bn = bn
        self._shape = None

    def _prepare(self, x: torch.Tensor, batches: int) -> None:
        bn = self.bn
        if x.shape[0] % batches:
            raise PleasHipError("bn_train_fold: %d samples do not split into %d batches" % (x.shape[0], batches))
        n, C = x.shape[0] // batches, x.shape[1]
        inner = math.prod(x.shape[2:])
        if n * inner <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        track = bn.track_running_stats and bn.running_mean is not None
        for t in (bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != x.device):
                raise PleasHipError("bn_train_fold: BatchNorm parameters / buffers must be contiguous fp32 on x's device")
        need = batches * int(_lib.lib().pleas_bn_train_ws_bytes(n, C))
        self._ws = torch.empty(need // 8, dtype=torch.float64, device=x.device)
        self._out = torch.empty(2, batches, C, dtype=torch.float32, device=x.device)
        ptr = lambda t: t.data_ptr() if t is not None else None
        self._tensors = self._addresses()      # compared per call: Module._apply swaps .data under the same objects
        self._args = (n, batches, C, inner, ptr(bn.weight), ptr(bn.bias), float(bn.eps),
                      ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                      ptr(bn.num_batches_tracked) if (track and bn.num_batches_tracked is not None) else None,
                      self._out[0].data_ptr(), self._out[1].data_ptr(), self._ws.data_ptr(), need)
        self._shape = (tuple(x.shape), x.device, batches)

    def _addresses(self) -> tuple:
        """What the cached arguments depend on: the addresses of the module's parameters / buffers (``.to()`` / ``.float()``
        swap ``.data`` under the SAME Parameter objects) and ``eps``."""
        bn = self.bn
        return tuple(t.data_ptr() if t is not None else 0 for t in
                     (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked)) + (float(bn.eps),)

    def __call__(self, x: torch.Tensor, batches: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
        """``batches`` > 1: ``x`` is that many batches back to back along dim 0; each is folded on its own samples, in
        order (running statistics and counter as after that many forwards); returns ``[batches, C]`` scale and shift."""
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() < 2:
            raise PleasHipError("bn_train_fold needs an fp32 [N, C, ...] tensor on the GPU")
        x = x.contiguous()
        bn = self.bn
        if self._shape != (tuple(x.shape), x.device, batches) or self._tensors != self._addresses():
            self._prepare(x, batches)
        n, nb, C, inner, w, b, eps, rm, rv, nbt, o0, o1, ws, need = self._args
        rc = _lib.lib().pleas_bn_train_fold_batches(x.data_ptr(), n, nb, C, inner, w, b, eps,
                                                    -1.0 if bn.momentum is None else float(bn.momentum), rm, rv, nbt, o0, o1,
                                                    ws, need, _stream())
        check(rc, "pleas_bn_train_fold_batches")
        return (self._out[0, 0], self._out[1, 0]) if batches == 1 else (self._out[0], self._out[1])


# ---------------------------------------------------------------------------------------- merge blocks
def merge_blocks(w1: torch.Tensor, w2: torch.Tensor, row_axis: int, row1: torch.Tensor, row2: torch.Tensor,
                 n_merged_rows: int, col1: Optional[torch.Tensor] = None, col2: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Block gather/average along ``row_axis`` (and ``row_axis + 1`` when column maps are given)."""
    _need_gpu(w1, w2)
    if w1.shape != w2.shape:
        raise PleasHipError("merge_blocks needs equal source shapes")
    w1, w2 = w1.contiguous(), w2.contiguous()
    shape = list(w1.shape)
    row_axis = row_axis % w1.dim()
    outer = math.prod(shape[:row_axis])
    rows_src = shape[row_axis]
    rows_out = int(row1.numel())
    if col1 is not None:
        cols_src, cols_out = shape[row_axis + 1], int(col1.numel())
        inner = math.prod(shape[row_axis + 2:])
        out_shape = shape[:row_axis] + [rows_out, cols_out] + shape[row_axis + 2:]
    else:
        cols_src = cols_out = 1
        inner = math.prod(shape[row_axis + 1:])
        out_shape = shape[:row_axis] + [rows_out] + shape[row_axis + 1:]
    maps = [row1, row2] + ([col1, col2] if col1 is not None else [])
    for m in maps:
        if not m.is_cuda or m.dtype != torch.int32 or not m.is_contiguous():
            raise PleasHipError("index maps must be contiguous int32 CUDA tensors")
    if out is None:
        out = torch.empty(out_shape, dtype=torch.float32, device=w1.device)
    elif list(out.shape) != out_shape or not out.is_contiguous():
        raise PleasHipError("out has the wrong shape")
    rc = _lib.lib().pleas_merge_blocks(
        w1.data_ptr(), w2.data_ptr(), out.data_ptr(), outer, rows_out, cols_out, inner, rows_src, cols_src,
        row1.data_ptr(), row2.data_ptr(), col1.data_ptr() if col1 is not None else None,
        col2.data_ptr() if col2 is not None else None, int(n_merged_rows), _stream())
    check(rc, "pleas_merge_blocks")
    return out


# ---------------------------------------------------------------------------------------- Adam / loss
def masked_adam(p: torch.Tensor, g: torch.Tensor, mask: Optional[torch.Tensor], m: torch.Tensor, v: torch.Tensor,
                lr: float, step: int, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    _need_gpu(p, g, m, v)
    n = p.numel()
    for t in (g, m, v) + ((mask,) if mask is not None else ()):
        if t.numel() != n or not t.is_contiguous():
            raise PleasHipError("masked_adam operands must be contiguous and equally sized")
    rc = _lib.lib().pleas_masked_adam(p.data_ptr(), g.data_ptr(), mask.data_ptr() if mask is not None else None,
                                      m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps, step, _stream())
    check(rc, "pleas_masked_adam")


def channel_sum(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """``out[c] = sum over n, p of x[n][c][p]`` (``x`` [N, C, ...] contiguous fp32): the bias gradient of a merged layer from
    its residual (reference: the bias node of ``total.backward()``, pleas_merging.py:287)."""
    _need_gpu(x, out)
    if not x.is_contiguous() or x.dim() < 2 or out.numel() != x.shape[1] or not out.is_contiguous():
        raise PleasHipError("channel_sum: x must be contiguous [N, C, ...] and out hold C floats")
    check(_lib.lib().pleas_channel_sum(x.data_ptr(), x.shape[0], x.shape[1], math.prod(x.shape[2:]), out.data_ptr(), _stream()),
          "pleas_channel_sum")
    return out


def sqerr(a: torch.Tensor, b: torch.Tensor, scale: float, out: torch.Tensor, accumulate: bool = False,
          diff: Optional[torch.Tensor] = None, dscale: float = 1.0) -> torch.Tensor:
    """out[0] (+)= scale * sum((a-b)^2); optionally diff = dscale * (a-b)."""
    _need_gpu(a, b, out)
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise PleasHipError("sqerr operands must be contiguous and equally shaped")
    n = a.numel()
    need = int(_lib.lib().pleas_sqerr_ws_bytes(n))
    ws = Workspace.get(a.device).reserve(need)
    rc = _lib.lib().pleas_sqerr(a.data_ptr(), b.data_ptr(), n, scale, int(bool(accumulate)), out.data_ptr(), dscale,
                                diff.data_ptr() if diff is not None else None, ws.data_ptr(), ws.numel(), _stream())
    check(rc, "pleas_sqerr")
    return out


# ---------------------------------------------------------------------------------------- PLeaS layer fitting
def target_residual(out: torch.Tensor, o1: torch.Tensor, o2: torch.Tensor, row1: torch.Tensor, row2: torch.Tensor,
                    n_merged: int, dscale: float, partials: torch.Tensor, resid: Optional[torch.Tensor] = None) -> int:
    """resid = dscale * (out - blockmerge(o1, o2)); per-workgroup sums of squares -> ``partials``.
    ``resid`` defaults to ``out`` (in place).  Returns the number of partials written."""
    _need_gpu(out, o1, o2, partials)
    if resid is None:
        resid = out
    if not (out.is_contiguous() and o1.is_contiguous() and o2.is_contiguous() and resid.is_contiguous()):
        raise PleasHipError("target_residual operands must be contiguous")
    N, C = out.shape[0], out.shape[1]
    HW = math.prod(out.shape[2:])
    if o1.shape != o2.shape or o1.shape[0] != N or math.prod(o1.shape[2:]) != HW or row1.numel() != C:
        raise PleasHipError("target_residual: shapes of merged output, source outputs and maps disagree")
    n = ctypes.c_int(0)
    rc = _lib.lib().pleas_target_residual(out.data_ptr(), o1.data_ptr(), o2.data_ptr(), row1.data_ptr(), row2.data_ptr(),
                                          int(n_merged), N, C, o1.shape[1], HW, dscale, resid.data_ptr(),
                                          partials.data_ptr(), ctypes.byref(n), _stream())
    check(rc, "pleas_target_residual")
    return n.value


def target_residual_max_partials() -> int:
    return int(_lib.lib().pleas_target_residual_max_partials())


def loss_final(partials: torch.Tensor, n_partials: torch.Tensor, scale: torch.Tensor, loss: torch.Tensor) -> None:
    """loss[l] = scale[l] * sum(partials[l, :n_partials[l]]) for all layers in one launch."""
    L, stride = partials.shape
    rc = _lib.lib().pleas_loss_final(partials.data_ptr(), n_partials.data_ptr(), scale.data_ptr(), stride, L,
                                     loss.data_ptr(), _stream())
    check(rc, "pleas_loss_final")


class MergeBatch:
    """Block gather / average of many tensors along one axis in ONE grouped launch (``pleas_merge_batch``): the merged
    inputs of all layers of a PLeaS update.  ``add`` returns the output tensor (filled at ``flush``)."""

    def __init__(self, device: torch.device):
        self.device = device
        self._keep: list = []
        self._geo: list = []
        self._arr = None
        self._ws = None
        self._fresh = 1

    def add(self, w1: torch.Tensor, w2: torch.Tensor, row_axis: int, row1: torch.Tensor, row2: torch.Tensor,
            n_merged_rows: int, out: Optional[torch.Tensor] = None, subsample: int = 1) -> torch.Tensor:
        """``subsample`` = s > 1 (``[N, C, H, W]`` sources merged along axis 1 only): the output keeps every s-th pixel of
        every s-th line, ``[N, rows, ceil(H / s), ceil(W / s)]`` -- what a 1x1 convolution with stride s reads."""
        if w1.shape != w2.shape or not w1.is_cuda or w1.dtype != torch.float32 or w2.dtype != torch.float32:
            raise PleasHipError("MergeBatch.add: fp32 CUDA sources of equal shape expected")
        w1, w2 = w1.contiguous(), w2.contiguous()
        shape = list(w1.shape)
        row_axis = row_axis % w1.dim()
        rows_out = int(row1.numel())
        want = shape[:row_axis] + [rows_out] + shape[row_axis + 1:]
        sub = (0, 0, 0)
        if subsample > 1:
            if w1.dim() != 4 or row_axis != 1:
                raise PleasHipError("MergeBatch.add: subsample needs [N, C, H, W] sources merged along axis 1")
            sub = (int(subsample), shape[2], shape[3])
            want = want[:2] + [-(-shape[2] // subsample), -(-shape[3] // subsample)]
        if out is None:
            out = torch.empty(want, dtype=torch.float32, device=w1.device)
        elif list(out.shape) != want or out.dtype != torch.float32 or not out.is_contiguous() or out.device != w1.device:
            raise PleasHipError("MergeBatch.add: out must be a contiguous fp32 tensor of shape %s" % (want,))
        self._keep.append((w1, w2, out, row1, row2))
        self._geo.append((math.prod(shape[:row_axis]), math.prod(want[row_axis + 1:]), rows_out, shape[row_axis],
                          int(n_merged_rows)) + sub)
        return out

    def flush(self) -> None:
        n = len(self._keep)
        if n == 0:
            return
        if self._arr is None or len(self._arr) != n:
            self._arr = (_lib.MergeItem * n)()
        for i, ((w1, w2, out, row1, row2), geo) in enumerate(zip(self._keep, self._geo)):
            a = self._arr[i]
            a.w1, a.w2, a.out, a.row1, a.row2 = w1.data_ptr(), w2.data_ptr(), out.data_ptr(), row1.data_ptr(), row2.data_ptr()
            a.outer, a.inner, a.rows_out, a.rows_src, a.n_merged, a.sub_stride, a.sub_h, a.sub_w = geo
        lib = _lib.lib()
        if self._ws is None:
            need = int(lib.pleas_merge_batch_ws_bytes(self._arr, n))
            if need == 0:
                raise PleasHipError("pleas_merge_batch_ws_bytes rejected the tensor list: %s" % lib.pleas_last_error().decode())
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._fresh = 1
        rc = lib.pleas_merge_batch(self._arr, n, self._ws.data_ptr(), self._ws.numel(), self._fresh, _stream())
        self._fresh = 0
        if rc == -12:
            self._ws = None
            self.flush()
            return
        check(rc, "pleas_merge_batch")
        self._keep.clear()
        self._geo.clear()



    def table(self):
        """numpy view of the item table of the latest ``flush`` (shares its memory): pointer fields may be rewritten before
        ``relaunch`` -- the per-update fast path of ``PleasFitter``, which keeps every other field as it is."""
        import numpy as np

        return None if self._arr is None else np.frombuffer(self._arr, dtype=np.dtype(type(self._arr[0])))

    def relaunch(self) -> None:
        """The launch of the latest ``flush`` again, with the table as it is now."""
        if self._keep or self._arr is None or self._ws is None:
            raise PleasHipError("MergeBatch.relaunch: nothing flushed yet, or tensors pending")
        check(_lib.lib().pleas_merge_batch(self._arr, len(self._arr), self._ws.data_ptr(), self._ws.numel(), 0, _stream()),
              "pleas_merge_batch")


def fwd_plan_lanes() -> dict:
    """Per tile form of the latest grouped forward: measured duration on its lane [ms], lane, work items
    (``pleas_fwd_plan_lanes``); ``state`` 2 = the lanes were dealt from those measurements."""
    ms, lane, items = (ctypes.c_double * 10)(), (ctypes.c_int * 10)(), (ctypes.c_int * 10)()
    state = _lib.lib().pleas_fwd_plan_lanes(ms, lane, items)
    return {"state": state, "forms": {f: {"ms": round(ms[f], 4), "lane": lane[f], "items": items[f]} for f in range(10) if items[f]}}


class FwdBatch:
    """Forward + target + residual + loss of all merged layers of one update in ONE grouped launch
    (``pleas_fwd_batch``).  ``add`` per layer, ``flush(loss)`` once per update."""

    def __init__(self, device: torch.device):
        self.device = device
        self._keep: list = []
        self._geo: list = []
        self._arr = None
        self._ws = None
        self._fresh = 1

    KPOS_MAJOR = 1   # w is [Cout][KH][KW][Cin] (needs Cin % 32 == 0)

    def add(self, ip, w, bias, o1, o2, row1, row2, n_merged: int, resid, dscale: float, loss_scale: float,
            kernel=(1, 1), stride: int = 1, pad: int = 0, flags: int = 0) -> None:
        for t in (ip, w, o1, o2, resid):
            if not t.is_contiguous():
                raise PleasHipError("FwdBatch.add: contiguous tensors expected")
        N, Cin = ip.shape[0], ip.shape[1]
        Hin, Win = (ip.shape[2], ip.shape[3]) if ip.dim() == 4 else (1, 1)
        self._keep.append((ip, w, bias, o1, o2, row1, row2, resid))
        self._geo.append((N, w.shape[0], Cin, Hin, Win, kernel[0], kernel[1], stride, pad, o1.shape[1], int(n_merged),
                          float(dscale), float(loss_scale), int(flags)))

    def flush(self, loss: torch.Tensor) -> None:
        n = len(self._keep)
        if n == 0:
            return
        if loss.numel() != n or not loss.is_contiguous():
            raise PleasHipError("FwdBatch.flush: loss must hold one float per layer")
        if self._arr is None or len(self._arr) != n:
            self._arr = (_lib.FwdLayer * n)()
        for i, (t, geo) in enumerate(zip(self._keep, self._geo)):
            a = self._arr[i]
            ip, w, bias, o1, o2, row1, row2, resid = t
            a.ip, a.w, a.bias = ip.data_ptr(), w.data_ptr(), (bias.data_ptr() if bias is not None else None)
            a.o1, a.o2, a.row1, a.row2, a.resid = o1.data_ptr(), o2.data_ptr(), row1.data_ptr(), row2.data_ptr(), resid.data_ptr()
            (a.N, a.Cout, a.Cin, a.Hin, a.Win, a.KH, a.KW, a.stride, a.pad, a.Csrc, a.n_merged, a.dscale,
             a.loss_scale, a.flags) = geo
        lib = _lib.lib()
        if self._ws is None:
            need = int(lib.pleas_fwd_batch_ws_bytes(self._arr, n))
            if need == 0:
                raise PleasHipError("pleas_fwd_batch_ws_bytes rejected the layer list: %s" % lib.pleas_last_error().decode())
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._fresh = 1
        rc = lib.pleas_fwd_batch(self._arr, n, loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(), self._fresh, _stream())
        self._fresh = 0
        if rc == -12:
            self._ws = None
            self.flush(loss)
            return
        check(rc, "pleas_fwd_batch")
        self._keep.clear()
        self._geo.clear()



    def table(self):
        """numpy view of the item table of the latest ``flush`` (shares its memory): pointer fields may be rewritten before
        ``relaunch`` -- the per-update fast path of ``PleasFitter``, which keeps every other field as it is."""
        import numpy as np

        return None if self._arr is None else np.frombuffer(self._arr, dtype=np.dtype(type(self._arr[0])))

    def relaunch(self, loss: torch.Tensor) -> None:
        """The launch of the latest ``flush`` again, with the table as it is now."""
        if self._keep or self._arr is None or self._ws is None or loss.numel() != len(self._arr):
            raise PleasHipError("FwdBatch.relaunch: nothing flushed yet, tensors pending, or loss of another length")
        check(_lib.lib().pleas_fwd_batch(self._arr, len(self._arr), loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(), 0,
                                         _stream()), "pleas_fwd_batch")


def conv2d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int, pad: int, kpos_major: bool = False,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.conv2d(x, w, bias, stride, pad)`` for dense, undilated, square geometries on the grouped forward's tile forms
    (``pleas_conv2d_fwd``): fp32 MFMA, a fixed summation order -- the same bits run after run, which MIOpen's 3 x 3
    kernels are not (they split K with atomics on small images / batches; tools/r05/probe_conv_classes.py).
    ``kpos_major``: ``w`` is ``[Cout][KH][KW][Cin]`` (needs Cin % 32 == 0; the flat-shift forms of stride-1 layers)."""
    _need_gpu(x, w)
    if x.dim() != 4 or w.dim() != 4 or not x.is_contiguous() or not w.is_contiguous():
        raise PleasHipError("conv2d: contiguous 4-D tensors expected")
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    KH, KW = (w.shape[1], w.shape[2]) if kpos_major else (w.shape[2], w.shape[3])
    if (w.shape[3] if kpos_major else w.shape[1]) != Cin:
        raise PleasHipError("conv2d: %d input channels vs a weight of %s" % (Cin, tuple(w.shape)))
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    check(_lib.lib().pleas_conv2d_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                      N, Cin, H, W, Cout, KH, KW, stride, pad, FwdBatch.KPOS_MAJOR if kpos_major else 0,
                                      _stream()), "pleas_conv2d_fwd")
    return out


def conv2d_bn_act(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int, pad: int, kpos_major: bool,
                  scale: torch.Tensor, shift: torch.Tensor, res: Optional[torch.Tensor], relu: bool):
    """``y = conv2d(x, w, bias)`` and ``z = bn_act(y, scale, shift, res, relu)`` from ONE launch (``pleas_conv2d_bn_act_fwd``):
    the activated image is written from the registers that hold y, bit for bit what ``bn_act`` makes of y.  Returns ``(y, z)``."""
    _need_gpu(x, w, scale, shift)
    if x.dim() != 4 or w.dim() != 4 or not x.is_contiguous() or not w.is_contiguous():
        raise PleasHipError("conv2d_bn_act: contiguous 4-D tensors expected")
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    KH, KW = (w.shape[1], w.shape[2]) if kpos_major else (w.shape[2], w.shape[3])
    if (w.shape[3] if kpos_major else w.shape[1]) != Cin:
        raise PleasHipError("conv2d_bn_act: %d input channels vs a weight of %s" % (Cin, tuple(w.shape)))
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if scale.numel() != Cout or shift.numel() != Cout or scale.dtype != torch.float32 or shift.dtype != torch.float32:
        raise PleasHipError("conv2d_bn_act: scale / shift must be fp32 vectors of %d channels" % Cout)
    if res is not None and (tuple(res.shape) != (N, Cout, Ho, Wo) or res.dtype != torch.float32 or not res.is_contiguous()
                            or res.device != x.device):
        raise PleasHipError("conv2d_bn_act: the identity must be a contiguous fp32 tensor shaped like the output")
    y = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    z = torch.empty_like(y)
    check(_lib.lib().pleas_conv2d_bn_act_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                             y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                             res.data_ptr() if res is not None else None, z.data_ptr(), 1 if relu else 0,
                                             N, Cin, H, W, Cout, KH, KW, stride, pad,
                                             FwdBatch.KPOS_MAJOR if kpos_major else 0, _stream()), "pleas_conv2d_bn_act_fwd")
    return y, z


class WgradBatch:
    """Weight gradients of all merged layers of one update in ONE grouped launch (``pleas_wgrad_batch``).
    ``add`` per layer (operands must stay unmodified until ``flush``), ``flush`` once per update."""

    def __init__(self, device: torch.device):
        self.device = device
        self._keep: list = []
        self._geo: list = []
        self._arr = None
        self._ws = None

    ACCUMULATE, KPOS_MAJOR = 1, 2

    def add(self, resid: torch.Tensor, ip: torch.Tensor, grad: torch.Tensor, kernel=(1, 1), stride: int = 1,
            pad: int = 0, flags: int = 0) -> None:
        if not (resid.is_contiguous() and ip.is_contiguous() and grad.is_contiguous()):
            raise PleasHipError("WgradBatch.add: contiguous tensors expected")
        N, Cout, Cin = resid.shape[0], resid.shape[1], ip.shape[1]
        Hin, Win = (ip.shape[2], ip.shape[3]) if ip.dim() == 4 else (1, 1)
        self._keep.append((resid, ip, grad))
        self._geo.append((N, Cout, Cin, Hin, Win, kernel[0], kernel[1], stride, pad, flags))

    def flush(self) -> None:
        n = len(self._keep)
        if n == 0:
            return
        if self._arr is None or len(self._arr) != n:
            self._arr = (_lib.WgradLayer * n)()
        for i, ((resid, ip, grad), geo) in enumerate(zip(self._keep, self._geo)):
            a = self._arr[i]
            a.resid, a.ip, a.grad = resid.data_ptr(), ip.data_ptr(), grad.data_ptr()
            a.N, a.Cout, a.Cin, a.Hin, a.Win, a.KH, a.KW, a.stride, a.pad, a.flags = geo
        lib = _lib.lib()
        if self._ws is None:
            need = int(lib.pleas_wgrad_batch_ws_bytes(self._arr, n))
            if need == 0:
                raise PleasHipError("pleas_wgrad_batch_ws_bytes rejected the layer list: %s" % lib.pleas_last_error().decode())
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._fresh = 1
        rc = lib.pleas_wgrad_batch(self._arr, n, self._ws.data_ptr(), self._ws.numel(), self._fresh, _stream())
        self._fresh = 0
        if rc == -12:
            self._ws = None
            self.flush()
            return
        check(rc, "pleas_wgrad_batch")
        self._keep.clear()
        self._geo.clear()



    def table(self):
        """numpy view of the item table of the latest ``flush`` (shares its memory): pointer fields may be rewritten before
        ``relaunch`` -- the per-update fast path of ``PleasFitter``, which keeps every other field as it is."""
        import numpy as np

        return None if self._arr is None else np.frombuffer(self._arr, dtype=np.dtype(type(self._arr[0])))

    def relaunch(self) -> None:
        """The launch of the latest ``flush`` again, with the table as it is now."""
        if self._keep or self._arr is None or self._ws is None:
            raise PleasHipError("WgradBatch.relaunch: nothing flushed yet, or tensors pending")
        check(_lib.lib().pleas_wgrad_batch(self._arr, len(self._arr), self._ws.data_ptr(), self._ws.numel(), 0, _stream()),
              "pleas_wgrad_batch")


class NormalEqBatch:
    """A_l += U_l^T U_l for all layers of one batch in ONE grouped launch (``pleas_normal_eq_accum``)."""

    def __init__(self, device: torch.device):
        self.device = device
        self._keep: list = []
        self._geo: list = []
        self._arr = None
        self._ws = None
        self._fresh = 1
        self._seen: dict = {}      # A's address -> (A, geometry): every matrix this object has accumulated into

    def add(self, ip: torch.Tensor, A: torch.Tensor, kernel=(1, 1), stride: int = 1, pad: int = 0) -> None:
        if not (ip.is_contiguous() and A.is_contiguous()):
            raise PleasHipError("NormalEqBatch.add: contiguous tensors expected")
        N, Cin = ip.shape[0], ip.shape[1]
        Hin, Win = (ip.shape[2], ip.shape[3]) if ip.dim() == 4 else (1, 1)
        K = kernel[0] * kernel[1] * Cin
        if tuple(A.shape) != (K, K):
            raise PleasHipError("NormalEqBatch.add: A must be (%d, %d)" % (K, K))
        self._keep.append((ip, A))
        self._geo.append((N, Cin, Hin, Win, kernel[0], kernel[1], stride, pad))
        self._seen[A.data_ptr()] = (A, self._geo[-1])

    def _layer_array(self, pairs):
        arr = (_lib.NeqLayer * len(pairs))()
        for a, (A, geo) in zip(arr, pairs):
            a.ip, a.A = 0, A.data_ptr()
            a.N, a.Cin, a.Hin, a.Win, a.KH, a.KW, a.stride, a.pad = geo
        return arr

    def seen(self) -> dict:
        """``{address of A: (A, geometry)}`` of every matrix this object has accumulated into."""
        return dict(self._seen)

    def finalize(self, pairs=None) -> None:
        """Fill the blocks ``flush`` leaves to the end (``pleas_normal_eq_finalize``: stride-1 k x k layers contract one
        block per lag class only).  Once, after the last batch -- and after the all-reduce of a multi-GPU run.
        ``pairs``: the ``(A, geometry)`` list to finalize when it is not this object's own -- a data-parallel rank that
        accumulated no batch still receives the all-reduced matrices and must complete ALL of them."""
        pairs = list(self._seen.values()) if pairs is None else list(pairs)
        if pairs:
            check(_lib.lib().pleas_normal_eq_finalize(self._layer_array(pairs), len(pairs), _stream()),
                  "pleas_normal_eq_finalize")

    def plan_info(self) -> dict:
        """Host-side facts about the grouped launch of the matrices seen so far (no GPU work)."""
        pairs = list(self._seen.values())
        info = (ctypes.c_double * 4)()
        if pairs:
            check(_lib.lib().pleas_normal_eq_plan_info(self._layer_array(pairs), len(pairs), info), "pleas_normal_eq_plan_info")
        return {"flops": info[0], "flops_executed": info[1], "items": int(info[2]), "blocks_to_finalize": int(info[3])}

    def flush(self) -> None:
        n = len(self._keep)
        if n == 0:
            return
        if self._arr is None or len(self._arr) != n:
            self._arr = (_lib.NeqLayer * n)()
        for i, ((ip, A), geo) in enumerate(zip(self._keep, self._geo)):
            a = self._arr[i]
            a.ip, a.A = ip.data_ptr(), A.data_ptr()
            a.N, a.Cin, a.Hin, a.Win, a.KH, a.KW, a.stride, a.pad = geo
        lib = _lib.lib()
        if self._ws is None:
            need = int(lib.pleas_normal_eq_ws_bytes(self._arr, n))
            if need == 0:
                raise PleasHipError("pleas_normal_eq_ws_bytes rejected the layer list: %s" % lib.pleas_last_error().decode())
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._fresh = 1
        rc = lib.pleas_normal_eq_accum(self._arr, n, self._ws.data_ptr(), self._ws.numel(), self._fresh, _stream())
        self._fresh = 0
        if rc == -12:
            self._ws = None
            self.flush()
            return
        check(rc, "pleas_normal_eq_accum")
        self._keep.clear()
        self._geo.clear()


def cholesky_solve_batched(As: Sequence[torch.Tensor], Bts: Sequence[torch.Tensor], ridge: float = 0.0) -> torch.Tensor:
    """In place: every row of ``Bts[p]`` (N_p x K_p) becomes the solution of ``x (A_p + ridge*mean(diag) I) = row``.
    ``As[p]`` (K_p x K_p, lower triangle read) is overwritten by its Cholesky factor.  Returns the device info vector."""
    k = len(As)
    if k == 0:
        return torch.zeros(0, dtype=torch.int32)
    _need_gpu(*As, *Bts)
    for a, b in zip(As, Bts):
        if a.dim() != 2 or a.shape[0] != a.shape[1] or b.dim() != 2 or b.shape[1] != a.shape[0] or not a.is_contiguous() \
                or not b.is_contiguous():
            raise PleasHipError("cholesky_solve_batched: A must be (K, K), Bt (N, K), both contiguous")
    info = torch.zeros(k, dtype=torch.int32, device=As[0].device)
    rc = _lib.lib().pleas_cholesky_solve_batched(
        (ctypes.c_void_p * k)(*[a.data_ptr() for a in As]), (ctypes.c_void_p * k)(*[b.data_ptr() for b in Bts]),
        (ctypes.c_int * k)(*[a.shape[0] for a in As]), (ctypes.c_int * k)(*[b.shape[0] for b in Bts]), k, float(ridge),
        info.data_ptr(), _stream())
    check(rc, "pleas_cholesky_solve_batched")
    return info


# ---------------------------------------------------------------------------------------- live kernel timing
def profile_enable(on: bool, skip: Sequence[str] = ()) -> None:
    """``skip``: kernel names left unrecorded (the two events per launch are not free for kernels that run
    hundreds of times per step, e.g. ``bn_act`` / ``merge_blocks``)."""
    mask = 0xFFFFFFFF
    for name in skip:
        mask &= ~(1 << _lib.PROF_KERNELS.index(name))
    _lib.lib().pleas_prof_select(mask)
    _lib.lib().pleas_prof_enable(int(bool(on)))


def profile_reset() -> None:
    _lib.lib().pleas_prof_reset()


def profile_collect() -> dict:
    """{kernel name: (launches, total_ms, algorithmic flops, algorithmic bytes)} since the last reset."""
    out = {}
    for kid, name in enumerate(_lib.PROF_KERNELS):
        n, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(_lib.lib().pleas_prof_collect(kid, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)),
              "pleas_prof_collect")
        if n.value:
            out[name] = (int(n.value), float(ms.value), float(fl.value), float(by.value))
    return out
