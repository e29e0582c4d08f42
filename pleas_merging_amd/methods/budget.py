"""Budget -> per-group merge ratios.  Host-side logic.

* ``count_linear_flops``  -- multiply-accumulate count of every Conv2d / Linear of a model and its description as
  terms ``(coefficient, group[, group])`` over the permutation groups (reference pleas/core/utils.py:558-617).
  Shapes come from a fake-tensor propagation (no real forward, unlike the reference).
* ``partial_merge_flops`` -- cost of the partially merged model under per-group ratios
  (reference pleas/methods/partial_matching.py:205-226).
* ``qp_ratios``           -- ratios that maximise ``sum w_k r_k`` under a relative FLOP budget.  The reference solves this
  non-convex program with Gurobi and returns RANDOM ratios when Gurobi is missing (:229-257, fallback :232-236);
  here it is solved without Gurobi (greedy ascent on value per FLOP + SLSQP polish, deterministic).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from ..core.compiler import trace_with_shapes
from ..core.utils import Axis, PermutationSpec
from .partial_matching import expand_ratios

Term = Tuple  # (coefficient, Axis) or (coefficient, Axis, Axis)


def count_linear_flops(spec: PermutationSpec, model: nn.Module, inputs_or_shapes) -> Tuple[int, List[Term]]:
    """``(flops, terms)``: per Conv2d / Linear ``coef = batch * prod(out spatial) * prod(kernel)`` (1 for Linear beyond
    the batch) times ``Cin * Cout``; axes that belong to a permutation group are replaced by the group key, sizes of
    the others are folded into the coefficient (reference utils.py:558-617)."""
    gm = trace_with_shapes(model, inputs_or_shapes)
    mods = dict(gm.named_modules())
    terms, sizes = [], {}
    for node in gm.graph.nodes:
        if node.op != "call_module":
            continue
        mod = mods[node.target]
        if not isinstance(mod, (nn.Conv2d, nn.Linear)):
            continue
        shape = node.meta["tensor_meta"].shape
        coeff = int(shape[0])
        if isinstance(mod, nn.Conv2d):
            coeff *= int(np.prod(shape[2:])) * int(np.prod(mod.kernel_size))
        ain, aout = Axis("%s.weight" % node.target, 1), Axis("%s.weight" % node.target, 0)
        sout, sin = int(mod.weight.shape[0]), int(mod.weight.shape[1])
        terms.append((coeff, ain, aout))
        if sizes.setdefault(ain, sin) != sin or sizes.setdefault(aout, sout) != sout:
            raise ValueError("inconsistent sizes for %s" % node.target)
    axis_keys = {ax: k for k, pg in spec.items() for ax in pg.state}
    flops, new_terms = 0, []
    for coef, *axes in terms:
        flops += coef * int(np.prod([sizes[a] for a in axes]))
        kept = []
        for a in axes:
            if a in axis_keys:
                kept.append(axis_keys[a])
            else:
                coef *= sizes[a]
        new_terms.append((coef, *kept))
    return flops, new_terms


def partial_merge_flops(spec: PermutationSpec, terms: Sequence[Term], ratios):
    """One-axis term: ``coef * size * (1 + r)``; two-axis term: ``coef * s1 * s2 * ((1 + r1)(1 + r2) - 2 r1 r2)``
    (a block that stays separate in both its input and output axes exists once per model, not crossed)."""
    ratios = expand_ratios(spec, ratios)
    total = 0
    for coef, *axes in terms:
        if len(axes) == 1:
            (a,) = axes
            total += coef * spec[a].size * (1 + ratios[a])
        elif len(axes) == 2:
            a, b = axes
            ra, rb = ratios[a], ratios[b]
            total += coef * spec[a].size * spec[b].size * ((1 + ra) * (1 + rb) - 2 * ra * rb)
        else:
            raise ValueError("terms carry one or two axes")
    return total


def _flops_model(spec: PermutationSpec, terms: Sequence[Term]):
    """F(r) = c0 + lin . r + r^T Q r  (Q symmetric, from the two-axis terms) as dense arrays over ``list(spec)``."""
    keys = list(spec.keys())
    pos = {k: i for i, k in enumerate(keys)}
    n = len(keys)
    c0, lin, Q = 0.0, np.zeros(n), np.zeros((n, n))
    for coef, *axes in terms:
        if len(axes) == 1:
            base = float(coef) * spec[axes[0]].size
            c0 += base
            lin[pos[axes[0]]] += base
        else:
            a, b = axes
            base = float(coef) * spec[a].size * spec[b].size
            c0 += base                       # (1 + ra)(1 + rb) - 2 ra rb = 1 + ra + rb - ra rb
            lin[pos[a]] += base
            lin[pos[b]] += base
            Q[pos[a], pos[b]] -= base / 2
            Q[pos[b], pos[a]] -= base / 2
    return keys, c0, lin, Q


def qp_ratios(spec: PermutationSpec, terms: Sequence[Term], flops_budget: float, obj_weights: Dict[Axis, float]) -> Dict[Axis, float]:
    """Ratios in [0, 1] maximising ``sum_k max(w_k, 1e-5) r_k`` subject to
    ``partial_merge_flops(r) / partial_merge_flops(0) <= flops_budget`` (reference objective and constraint,
    partial_matching.py:239-252), solved without Gurobi.  Deterministic; ``{key: ratio}`` like the reference."""
    from scipy.optimize import minimize

    keys, c0, lin, Q = _flops_model(spec, terms)
    n = len(keys)
    w = np.array([max(float(obj_weights[k]), 1e-5) for k in keys])
    cap = float(flops_budget) * c0
    F = lambda r: c0 + lin @ r + r @ Q @ r
    dF = lambda r: lin + 2 * Q @ r
    if F(np.ones(n)) <= cap:
        return {k: 1.0 for k in keys}
    if cap <= c0:
        return {k: 0.0 for k in keys}
    # greedy ascent: raise the coordinate with the best value per marginal FLOP in small steps until the budget binds
    r = np.zeros(n)
    step = 1.0 / 64
    for _ in range(64 * n + 1):
        grad = dF(r)
        score = np.where(r < 1.0 - 1e-12, w / np.maximum(grad, 1e-12), -np.inf)
        order = np.argsort(-score, kind="stable")
        moved = False
        for i in order:
            if not np.isfinite(score[i]):
                break
            trial = r.copy()
            trial[i] = min(1.0, r[i] + step)
            if F(trial) <= cap:
                r, moved = trial, True
                break
            lo, hi = r[i], trial[i]          # the budget binds inside this step: bisect
            for _ in range(40):
                mid = (lo + hi) / 2
                trial[i] = mid
                lo, hi = (mid, hi) if F(trial) <= cap else (lo, mid)
            if lo > r[i] + 1e-9:
                r = r.copy()
                r[i] = lo
                moved = True
                break
        if not moved:
            break
    best, best_val = r, float(w @ r)
    res = minimize(lambda v: -(w @ v), r, jac=lambda v: -w, bounds=[(0.0, 1.0)] * n, method="SLSQP",
                   constraints=[{"type": "ineq", "fun": lambda v: (cap - F(v)) / c0, "jac": lambda v: -dF(v) / c0}],
                   options={"maxiter": 200, "ftol": 1e-10})
    if res.success:
        cand = np.clip(res.x, 0.0, 1.0)
        if F(cand) > cap:                   # pull a slightly infeasible polish back onto the budget along the ray to r
            lo, hi = 0.0, 1.0
            for _ in range(50):
                mid = (lo + hi) / 2
                lo, hi = (mid, hi) if F(r + mid * (cand - r)) <= cap else (lo, mid)
            cand = r + lo * (cand - r)
        if float(w @ cand) > best_val + 1e-12:
            best = cand
    return {k: float(min(max(v, 0.0), 1.0)) for k, v in zip(keys, best)}
