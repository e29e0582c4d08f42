"""Weight matching (Git Re-Basin coordinate descent) on the HIP kernels.

Drop-in for the reference's ``pleas/methods/weight_matching.py:22-95``: the groups are visited in a seeded random order;
a visit scores group ``p`` by ``A = sum over its state axes of Wa . Wb^T``, solves the LAP on ``A``, and permutes model B's
tensors along ``p``; sweeps repeat until one of them improves no group.

MI355X design.  Consecutive visits that do not depend on each other -- group q's score reads no tensor that an earlier
visit of the run permutes; in a ResNet that is most of them: an inner group shares tensors only with its sibling in the
block and with its stage's residual stream -- are one BATCH: their score matrices come from before any of the batch's
permutations, exactly what the sequential sweep computes for each of them.  A batch is three enqueues and NO host
synchronisation:
  * ONE grouped contraction (``pleas_gram_batch``, inner-product epilogue) over ALL state axes of ALL its groups -- any
    axis of a weight tensor is a ``[B][C][HW]`` view, so nothing is moved or reshaped (round 2 launched
    ``pleas_gram_accum`` once per axis: up to 144 launches for ResNet-101's layer3 residual group);
  * ONE ``pleas_lsap_batched`` launch over its groups' matrices, whose assignments STAY on the device;
  * the gathers of ``apply_perm`` along those device indices, plus three scalars per group (gain, progress, norm)
    appended to a device log.
The host reads the log ONCE per sweep (the only thing it needs from a sweep is "did any group improve") and the
permutations once at the end.  With ``verbose`` the per-visit lines are printed from that log after the sweep, in visit
order.  Callers that hold HOST state dicts pass explicit ``cross_weights`` / ``lsa_solver`` callables (the reference's plug
points, BASELINE.json configs[0]) and take the generic visit below; device tensors never fall back to it.
"""
from __future__ import annotations

from copy import copy, deepcopy
from typing import Callable, Dict, List, Sequence, Tuple, Union

import torch

from ..core.solvers import hip_solve_lsa
from ..core.utils import Axis, Permutation, PermutationSpec, StateDict, apply_perm, make_identity_perm
from ..hip_ops import cross_features_inner_product


class _Visitor:
    """State of one ``weight_matching`` call: the two lists of state dicts, which axes score a group, which visits of a
    sweep can share their launches, and how groups are scored and solved.  ``visit(ps)`` returns per group ``(A, new,
    report)``: score matrix, assignment, and a 3-vector on ``A``'s device -- gain of ``new`` over the identity, whether
    that counts as progress, ``|A|`` -- that the caller reads later."""

    def __init__(self, spec: PermutationSpec, state_as: List[StateDict], state_bs: List[StateDict], skip: Tuple[str, ...],
                 skip_missing: bool, cross_weights: Callable, lsa_solver: Callable, batch_runs: bool = False):
        self.spec, self.state_as, self.state_bs = spec, state_as, state_bs
        self.batch_runs = batch_runs      # tests: form independent runs on the plug-point path too (the default there is one
                                          # visit at a time, as the reference)
        self.cross_weights, self.lsa_solver = cross_weights, lsa_solver
        self.device = next(iter(state_as[0].values())).device
        # on the device: default callables = the grouped launch + the batched LAP kernel
        self.grouped = cross_weights is cross_features_inner_product
        self.device_lap = lsa_solver is hip_solve_lsa
        if self.grouped or self.device_lap:
            from .. import hip_ops

            self.ops = hip_ops
            if self.device.type != "cuda":
                raise hip_ops.PleasHipError(
                    "weight_matching: state dicts are on %s; the default HIP cross_weights / lsa_solver need GPU tensors "
                    "(pass explicit cross_weights= / lsa_solver= callables to run elsewhere)" % self.device)
        # the axes that score a group, per group, in the group's own iteration order (reference :65-75)
        self.axes: Dict[Axis, List[Tuple[int, Axis]]] = {}
        for p, group in spec.items():
            use = []
            for ax in group.state:
                if ax.key.endswith(skip):
                    continue
                for m, (sa, sb) in enumerate(zip(state_as, state_bs)):
                    if skip_missing and not (ax.key in sa and ax.key in sb):
                        continue
                    use.append((m, ax))
            self.axes[p] = use
        self._batches: Dict[Tuple[Axis, ...], object] = {}      # batch composition -> (score matrices, their GramBatch)

    def _score(self, p: Axis) -> torch.Tensor:
        """Generic plug-point path: ``A = sum of cross_weights(...)`` for one group."""
        n = self.spec[p].size
        A = torch.zeros(n, n, device=self.device)
        for m, ax in self.axes[p]:
            A.add_(self.cross_weights(self.state_as[m][ax.key], self.state_bs[m][ax.key], ax.axis))
        return A

    def _score_grouped(self, ps: Tuple[Axis, ...]) -> List[torch.Tensor]:
        """Score matrices of all groups of a batch from ONE grouped launch (cached per batch composition)."""
        hit = self._batches.get(ps)
        if hit is None:
            mats = [torch.empty(self.spec[p].size, self.spec[p].size, dtype=torch.float32, device=self.device) for p in ps]
            if len(self._batches) >= 256:       # compositions change from sweep to sweep: keep the cache bounded
                self._batches.clear()
            hit = self._batches[ps] = (mats, self.ops.GramBatch(mats, self.ops.EPI_INNER))
        mats, batch = hit
        for g, p in enumerate(ps):
            for m, ax in self.axes[p]:
                batch.add(self.state_as[m][ax.key], self.state_bs[m][ax.key], ax.axis, g)
            if not self.axes[p]:
                mats[g].zero_()
        batch.flush(accumulate=False)
        return mats

    def independent_run(self, order: List[Axis], start: int) -> int:
        """End (exclusive) of the longest run ``order[start:end]`` whose visits commute with the sequential sweep: no
        group of the run scores a tensor that an EARLIER group of the run permutes (reference :59-91 visits one by one;
        a visit permutes every state tensor of its group, :88)."""
        if not ((self.grouped and self.device_lap) or self.batch_runs):
            return start + 1
        dirty: set = set()
        end = start
        while end < len(order):
            q = order[end]
            if end > start and any((m, ax.key) in dirty for m, ax in self.axes[q]):
                break
            for ax in self.spec[q].state:
                for m in range(len(self.state_bs)):
                    dirty.add((m, ax.key))
            end += 1
        return end

    def visit(self, ps: Tuple[Axis, ...]):
        """One batch of mutually independent visits: ``[(A, new, report)]`` per group, in order."""
        if self.grouped and self.device_lap:
            mats = self._score_grouped(ps)
            news = self.ops.solve_lsa_batched(mats, maximize=True)
        else:
            mats = [self._score_grouped((p,))[0] if self.grouped else self._score(p) for p in ps]
            news = [self.ops.solve_lsa_batched([A], maximize=True)[0] if self.device_lap else self.lsa_solver(A) for A in mats]
        out = []
        for p, A, new in zip(ps, mats, news):
            pick = new.to(A.device)
            before, after = A.diag().sum(), A[torch.arange(A.shape[0], device=A.device), pick].sum()     # reference :80
            # what the caller reads once per sweep: gain, "improved" by the reference's fp32 test (:81), the matrix's norm
            report = torch.stack([after - before, (after > before + 1e-12).to(A.dtype), A.norm()])
            # the matrix goes to the caller (costs[p]); the grouped path reuses its buffer when the composition recurs
            out.append((A.clone() if self.grouped else A, new, report, pick))
        for p, (_, _, _, pick) in zip(ps, out):       # all scores of the batch are taken: now the permutations
            for sb in self.state_bs:
                apply_perm({p: pick}, self.spec, sb, inplace=True)
        return [(A, new, report) for A, new, report, _ in out]


def weight_matching(
    spec: PermutationSpec,
    state_as: Union[StateDict, Sequence[StateDict]],
    state_bs: Union[StateDict, Sequence[StateDict]],
    max_iter=100,
    init_perm=None,
    inplace=False,
    skip_suffixes=("running_mean", "running_var"),
    skip_missing=True,
    lsa_solver=hip_solve_lsa,
    cross_weights=cross_features_inner_product,
    verbose=True,
    seed=0,
    return_costs=False,
) -> Permutation:
    """Same signature, defaults and return value as the reference (``perm`` or ``(perm, costs)``; ``costs[p]`` is the
    score matrix of group ``p``'s LAST visit, in already-permuted-B coordinates -- reference :85-88)."""
    state_as = [state_as] if isinstance(state_as, dict) else list(state_as)
    state_bs = [state_bs] if isinstance(state_bs, dict) else list(state_bs)
    assert len(state_as) == len(state_bs)
    if not inplace:
        state_bs = [copy(sb) for sb in state_bs]
    perm = make_identity_perm(spec) if init_perm is None else deepcopy(init_perm)
    if init_perm is not None:
        for sb in state_bs:
            apply_perm(init_perm, spec, sb, inplace=True)

    names = list(perm.keys())
    rng = torch.Generator()
    rng.manual_seed(seed)
    costs: Dict[Axis, torch.Tensor] = {}
    with torch.no_grad():
        walker = _Visitor(spec, state_as, state_bs, tuple(skip_suffixes), skip_missing, cross_weights, lsa_solver)
        composed = {p: v.to(walker.device) for p, v in perm.items()}      # perm[p] <- perm[p][new], where `new` lives
        for sweep in range(max_iter):
            order = [names[i] for i in torch.randperm(len(names), generator=rng)]
            log = []
            at = 0
            while at < len(order):
                end = walker.independent_run(order, at)
                for p, (A, new, report) in zip(order[at:end], walker.visit(tuple(order[at:end]))):
                    costs[p] = A
                    composed[p] = composed[p][new.to(composed[p].device)]
                    log.append(report)
                at = end
            rows = torch.stack(log).cpu()            # ONE read per sweep
            assert bool((rows[:, 2] > 0).all()), "weight_matching: a group's score matrix is zero"      # reference :77
            if verbose:
                for p, gain in zip(order, rows[:, 0].tolist()):
                    print("%d/%s:%d: %s" % (sweep, p.key, p.axis, gain))
            if not bool((rows[:, 1] > 0).any()):
                break
    perm = {p: v.cpu() for p, v in composed.items()}
    return (perm, costs) if return_costs else perm
