"""Evaluation helpers for merged models with separate heads: the first consumers of
the block layout ``[merged | separate-of-model-1 | separate-of-model-2]`` that ``partial_merge`` gives every axis.

Reference: pleas/methods/pleas_merging.py:408-496 (``get_fc_perm``, ``permute_final_features``, ``eval_perm_model``)
and :575-586 (``eval_whole_model``).  Accuracy is counted directly (torchmetrics is not a dependency here).
"""
from __future__ import annotations

from typing import Iterable

import torch
from torch import nn

from ..core.utils import Axis, Permutation, PermutationSpec
from .partial_matching import get_blocks


def get_fc_perm(perm: Permutation, spec: PermutationSpec, costs, budget_ratios):
    """Blocks ``(merged idx 1, merged idx 2, separate idx 1, separate idx 2)`` of the group that feeds the classifier
    (the one whose state holds ``fc.weight:1``).  Reference :408-433."""
    blocks = get_blocks(spec, perm, costs, budget_ratios, False)
    for key, group in spec.items():
        if Axis("fc.weight", 1) in group.state:
            return blocks[key]
    raise ValueError("No fc perm found")


def permute_final_features(features: torch.Tensor, fc_perm, idx: int) -> torch.Tensor:
    """Backbone features of the merged model ``[N, n_merged + 2 n_separate]`` -> the feature order source model
    ``idx`` (0 or 1) was trained with, so that its own classifier applies.  Reference :436-466."""
    b1, b2, b1c, b2c = fc_perm
    ni, mi = len(b1), len(b1c)
    merged = features[:, :ni]
    if idx == 0:
        own, order = features[:, ni:ni + mi], torch.cat([b1, b1c], 0)
    else:
        own, order = features[:, ni + mi:ni + 2 * mi], torch.cat([b2, b2c], 0)
    return torch.cat([merged, own], 1)[:, torch.argsort(order).to(features.device)]


@torch.no_grad()
def eval_perm_model(model: nn.Module, fc: nn.Module, dataloader: Iterable, num_classes: int, fc_perm, idx: int) -> torch.Tensor:
    """Top-1 accuracy of the merged backbone under the classifier of source model ``idx``.  Reference :469-496
    (``num_classes`` is kept for the signature; the count does not need it)."""
    device = next(iter(model.parameters())).device
    model.eval()
    hit = torch.zeros((), dtype=torch.long, device=device)
    seen = 0
    for x, y in dataloader:
        x, y = x.to(device), y.to(device)
        logits = fc(permute_final_features(model(x), fc_perm, idx))
        hit += (logits.argmax(1) == y).sum()
        seen += int(y.numel())
    return hit.float() / max(seen, 1)


@torch.no_grad()
def eval_whole_model(model: nn.Module, dataloader: Iterable, num_classes: int) -> torch.Tensor:
    """Top-1 accuracy of a model with its own head.  Reference :575-586."""
    device = next(iter(model.parameters())).device
    model.eval()
    hit, seen = 0, 0
    for x, y in dataloader:
        pred = model(x.to(device)).argmax(1).cpu()
        hit += int((pred == y.cpu()).sum())
        seen += int(y.numel())
    acc = torch.tensor(hit / max(seen, 1))
    print(acc)
    return acc


def train_eval_linear_probe(model: nn.Module, train_dataloader, test_dataloader, num_classes: int, wandb_run, dataset_name: str,
                            lr: float = 1e-3, epochs: int = 10, device=None) -> nn.Module:
    """Linear probe on a frozen (merged) backbone, as the different-label-space driver evaluates merged models
    (reference :499-570; run_torchvision.py:276-290).  Same recipe: Adam(``lr``) on a fresh ``Linear`` head, cosine
    schedule over ``epochs * len(train_dataloader)`` steps down to ``lr / 10``, cross entropy, backbone in eval mode
    under ``no_grad``; per-epoch train accuracy / loss and the final test accuracy go to ``wandb_run.log`` under the
    reference's keys (``wandb_run=None`` skips logging).  Returns the trained head.  The backbone stays where it is
    (``device`` defaults to its device) -- the reference hard-codes ``.cuda()`` and a 224x224 probe input; here the
    feature width is read from the first training batch."""
    if device is None:
        device = next(iter(model.parameters())).device
    model.eval()
    n_batches = len(train_dataloader)
    fc = opt = sched = None
    loss_fn = nn.CrossEntropyLoss()
    log = wandb_run.log if wandb_run is not None else (lambda _metrics: None)
    for epoch in range(epochs):
        hit = torch.zeros((), dtype=torch.long, device=device)
        seen, total, loss = 0, 0.0, None
        for x, y in train_dataloader:
            x, y = x.to(device), y.to(device)
            with torch.no_grad():
                feats = model(x)
            if fc is None:
                fc = nn.Linear(feats.shape[-1], num_classes).to(device)
                opt = torch.optim.Adam(fc.parameters(), lr=lr)
                sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, epochs * n_batches, eta_min=lr / 10)
            fc.train()
            logits = fc(feats)
            loss = loss_fn(logits, y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            sched.step()
            hit += (logits.argmax(1) == y).sum()
            seen += int(y.numel())
            total += float(loss)
        log({"%s_linear_probe_train_acc" % dataset_name: float(hit) / max(seen, 1),
             "%s_linear_probe_train_loss" % dataset_name: float(loss) if loss is not None else float("nan"),
             "epoch": epoch, "%s_total_loss" % dataset_name: total / max(n_batches, 1)})
    if fc is None:
        raise ValueError("train_eval_linear_probe: empty training loader")
    fc.eval()
    hit, seen = torch.zeros((), dtype=torch.long, device=device), 0
    with torch.no_grad():
        for x, y in test_dataloader:
            x, y = x.to(device), y.to(device)
            hit += (fc(model(x)).argmax(1) == y).sum()
            seen += int(y.numel())
    log({"%s_linear_probe_acc" % dataset_name: float(hit) / max(seen, 1)})
    return fc
