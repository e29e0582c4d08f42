"""Callers and data formats either side of the hot path ("next" rows).

Host-side logic only; the forwards run under PyTorch-ROCm like everything else outside the kernels.

* ``reset_bn_stats``   -- BN-statistics reset pass that every driver runs right after ``train``
                          (experiments/shared_label_space/run_domainnet.py:327-341,
                          experiments/different_label_space/run_torchvision.py:268-274)
* ``zip_ratios``       -- budget ratio -> per-group merge ratios of the "zip" strategy
                          (run_domainnet.py:34-56; keyed on ``Axis.key`` -- the published function
                          calls ``str.startswith`` on ``Axis`` keys and raises)
* ``save_matching`` / ``load_matching`` -- perm + costs on disk so LAP / merge / PLeaS can be re-run
                          without the forwards (no reference counterpart; resumability)
* ``load_checkpoint``  -- input checkpoints as the drivers read them: a raw state dict or ``['model']``
                          (run_domainnet.py:190-193, run_torchvision.py:176-182)

The other 8(f) rows live next door: ``budget.py`` (FLOP model, Gurobi-free ``qp_ratios``) and ``evaluation.py``
(``get_fc_perm`` / ``permute_final_features`` / ``eval_perm_model``).
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence

import torch
from torch import nn

from ..core.utils import Axis, Permutation, PermutationSpec


def _bn_reset_runner(model: nn.Module, fused: bool):
    """What the pass forwards: on the GPU, an fx copy of ``model`` (same submodules, same buffers) whose train-mode
    ``BatchNorm2d -> [+ identity] -> [ReLU]`` chains are ``pleas_bn_train_fold`` + ONE ``pleas_bn_act`` pass each -- the
    statistics kernel moves the module's running statistics and counter as the module would; on the CPU, or when the
    model cannot be traced, the model itself (vendor modules)."""
    if fused and next(iter(model.parameters())).is_cuda:
        from .source_forward import fuse_bn_act

        return fuse_bn_act(model, train_stats=True) or model
    return model


@torch.no_grad()
def reset_bn_stats(model: nn.Module, dataloader: Iterable, num_batches: int = 101, device=None, shard: bool = True,
                   fused: bool = True, batches_per_forward=None) -> nn.Module:
    """Recompute BatchNorm running statistics of a (merged) model on data.

    ``fused=True`` (default): a model on the GPU is forwarded through the HIP BatchNorm path (``_bn_reset_runner``);
    ``fused=False`` runs the vendor modules (what the reference's loop does, kept for A/B).  ``batches_per_forward``
    (HIP path, one process): consecutive equal batches per forward, each still normalised and counted on its own.

    Same procedure as the reference drivers: ``model.train()``, ``reset_running_stats()`` on every
    ``BatchNorm2d``, ``num_batches`` forward passes without gradients (the drivers break after 101),
    default momentum.  Leaves the model in train mode like the reference does; callers ``eval()`` next.

    Data parallel (``torch.distributed`` initialised, one process per GPU): rank r forwards batches ``b % world == r``.
    A train-mode forward normalises with the batch's own statistics, so batches are independent, and the running
    statistics after n sequential updates are the fixed linear combination
    ``(1-m)^n r0 + sum_b m (1-m)^(n-1-b) stat_b``: every rank accumulates its share of the sum and ONE all-reduce of
    a flat buffer per statistic gives every rank the sequential result (``momentum=None``, the cumulative average: the
    plain mean of the batches' statistics).
    """
    from .activation_matching import _dist_info

    if device is None:
        device = next(iter(model.parameters())).device
    rank, world = _dist_info() if shard else (0, 1)
    model.train()
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    for m in bns:
        m.reset_running_stats()
    runner = _bn_reset_runner(model, fused)        # built AFTER model.train(): the rewrite folds what is in train mode
    if world == 1:
        # HIP path: consecutive batches of equal shape share a forward (the vendor convolutions run faster per sample on
        # 64 samples than on 16) -- every BatchNorm still folds each batch on its own samples, in order, in one launch, so
        # statistics, running statistics and counters are those of one forward per batch.  Default: forwards of up to 64
        # samples, at most 4 batches.
        together = getattr(runner, "all_batch_statistics_folded", False) and hasattr(runner, "batches_per_forward")
        per, run = (None if batches_per_forward is None else max(1, int(batches_per_forward))), []

        def forward(xs):
            if len(xs) == 1:
                runner(xs[0])
                return
            runner.batches_per_forward.parts = len(xs)
            try:
                runner(torch.cat(xs, 0))
            finally:
                runner.batches_per_forward.parts = 1

        for i, batch in enumerate(dataloader):
            if i >= num_batches:
                break
            x = batch[0].to(device).float()
            if not together:
                runner(x)
                continue
            if per is None:
                per = max(1, min(4, 64 // max(1, int(x.shape[0]))))
            if run and (x.shape != run[0].shape or len(run) == per):
                forward(run)
                run = []
            run.append(x)
        if run:
            forward(run)
        return model

    import torch.distributed as dist

    tracked = [m for m in bns if m.track_running_stats and m.running_mean is not None]
    saved = [m.momentum for m in tracked]      # None = cumulative average: the running statistics are the plain mean
    sizes = [m.num_features for m in tracked]
    acc = torch.zeros(2, sum(sizes), dtype=torch.float64, device=device)     # weighted sums of batch means / variances
    n = 0
    try:
        for m in tracked:
            m.momentum = 1.0            # after a forward, running_* hold exactly this batch's statistics
        for i, batch in enumerate(dataloader):
            if i >= num_batches:
                break
            n = i + 1
            if i % world != rank:
                continue
            runner(batch[0].to(device).float())
            off = 0
            for m, mom, c in zip(tracked, saved, sizes):
                wgt = 1.0 if mom is None else (1.0 - mom) ** (-i)            # common factor m (1-m)^(n-1) applied at the end
                acc[0, off:off + c].add_(m.running_mean.double(), alpha=wgt)
                acc[1, off:off + c].add_(m.running_var.double(), alpha=wgt)
                off += c
    finally:
        for m, mom in zip(tracked, saved):
            m.momentum = mom
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    off = 0
    for m, mom, c in zip(tracked, saved, sizes):
        scale, keep = (1.0 / max(n, 1), 0.0) if mom is None else (mom * (1.0 - mom) ** (n - 1), (1.0 - mom) ** n)
        m.running_mean.copy_((acc[0, off:off + c] * scale).float())           # r0 = 0
        m.running_var.copy_((acc[1, off:off + c] * scale + keep).float())     # r0 = 1
        m.num_batches_tracked.fill_(n)
        off += c
    return model


def zip_ratios(spec: PermutationSpec, budget_ratio: float,
               base_budget_ratios: Sequence[float] = (1.0, 1.2, 1.55, 1.8, 2.0)) -> Dict[Axis, float]:
    """Per-group ratios of the zip strategy: budgets ``base_budget_ratios[i]`` keep stages ``> 4 - i``
    separate (ratio 1.0) and merge everything else (ratio 0.0); non-``layerN`` groups (the stem) are
    always merged.  (Reference: run_domainnet.py:34-56; defaults: merge_configs.py:25-27 for rn50/rn101.)"""
    last_merged = {b: 4 - i for i, b in enumerate(base_budget_ratios)}
    if budget_ratio not in last_merged:
        raise KeyError("budget_ratio %r is not one of %r" % (budget_ratio, tuple(base_budget_ratios)))
    out: Dict[Axis, float] = {}
    for key in spec:
        name = key.key
        if name.startswith("layer"):
            stage = int(name.split(".")[0][len("layer"):])
            out[key] = 0.0 if stage <= last_merged[budget_ratio] else 1.0
        else:
            out[key] = 0.0
    return out


def save_matching(path: str, perm: Permutation, costs: Dict[Axis, torch.Tensor]) -> None:
    """``torch.save`` of ``{"perm": {str(axis): LongTensor}, "costs": {str(axis): FloatTensor (CPU)}}``."""
    torch.save({"perm": {str(k): v.cpu() for k, v in perm.items()},
                "costs": {str(k): v.detach().cpu() for k, v in costs.items()}}, path)


def load_matching(path: str, device="cpu"):
    """Inverse of :func:`save_matching`: ``(perm, costs)`` keyed by ``Axis``; costs moved to ``device``."""
    blob = torch.load(path, map_location="cpu")
    perm = {Axis.parse(k): v for k, v in blob["perm"].items()}
    costs = {Axis.parse(k): v.to(device) for k, v in blob["costs"].items()}
    return perm, costs


def load_checkpoint(model: nn.Module, path: str) -> nn.Module:
    """Load a driver-style checkpoint: a raw state dict, or a dict with the weights under ``'model'``."""
    blob = torch.load(path, map_location="cpu")
    if isinstance(blob, dict) and "model" in blob and all(not torch.is_tensor(v) for k, v in blob.items() if k != "model"):
        blob = blob["model"]
    model.load_state_dict(blob)
    return model
