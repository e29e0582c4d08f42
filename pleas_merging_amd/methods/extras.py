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
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence

import torch
from torch import nn

from ..core.utils import Axis, Permutation, PermutationSpec


@torch.no_grad()
def reset_bn_stats(model: nn.Module, dataloader: Iterable, num_batches: int = 101, device=None) -> nn.Module:
    """Recompute BatchNorm running statistics of a (merged) model on data.

    Same procedure as the reference drivers: ``model.train()``, ``reset_running_stats()`` on every
    ``BatchNorm2d``, ``num_batches`` forward passes without gradients (the drivers break after 101),
    default momentum.  Leaves the model in train mode like the reference does; callers ``eval()`` next.
    """
    if device is None:
        device = next(iter(model.parameters())).device
    model.train()
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.reset_running_stats()
    for i, batch in enumerate(dataloader):
        if i >= num_batches:
            break
        model(batch[0].to(device).float())
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
