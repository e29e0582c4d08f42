"""Weight matching (Git Re-Basin coordinate descent) on the HIP kernels.

Drop-in for the reference's ``pleas/methods/weight_matching.py:22-95``: the groups are visited in a seeded random order;
a visit scores group ``p`` by ``A = sum over its state axes of Wa . Wb^T``, solves the LAP on ``A``, and permutes model B's
tensors along ``p``; sweeps repeat until one of them improves no group.

MI355X design.  A visit is three enqueues and NO host synchronisation:
  * ONE grouped contraction (``pleas_gram_batch``, inner-product epilogue) over ALL state axes of the group -- any axis of
    a weight tensor is a ``[B][C][HW]`` view, so nothing is moved or reshaped (round 2 launched ``pleas_gram_accum`` once
    per axis: up to 144 launches for ResNet-101's layer3 residual group);
  * ONE ``pleas_lsap_batched`` launch whose assignment STAYS on the device;
  * the gathers of ``apply_perm`` along that device index, plus two scalars (score before / after) appended to a device
    log.
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
    """State of one ``weight_matching`` call: the two lists of state dicts, which axes score a group, and how a group is
    scored and solved.  ``visit(p)`` returns ``(A, new, report)``: score matrix, assignment, and a 3-vector on ``A``'s
    device -- gain of ``new`` over the identity, whether that counts as progress, ``|A|`` -- that the caller reads later."""

    def __init__(self, spec: PermutationSpec, state_as: List[StateDict], state_bs: List[StateDict], skip: Tuple[str, ...],
                 skip_missing: bool, cross_weights: Callable, lsa_solver: Callable):
        self.spec, self.state_as, self.state_bs = spec, state_as, state_bs
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
        self._batches: Dict[Axis, object] = {}      # group -> (score matrix, its GramBatch), kept across sweeps

    def _score(self, p: Axis) -> torch.Tensor:
        n = self.spec[p].size
        if not self.grouped:
            A = torch.zeros(n, n, device=self.device)
            for m, ax in self.axes[p]:
                A.add_(self.cross_weights(self.state_as[m][ax.key], self.state_bs[m][ax.key], ax.axis))
            return A
        hit = self._batches.get(p)
        if hit is None:
            A = torch.empty(n, n, dtype=torch.float32, device=self.device)
            hit = self._batches[p] = (A, self.ops.GramBatch([A], self.ops.EPI_INNER))
        A, batch = hit
        for m, ax in self.axes[p]:
            batch.add(self.state_as[m][ax.key], self.state_bs[m][ax.key], ax.axis, 0)
        if not self.axes[p]:
            A.zero_()
        batch.flush(accumulate=False)
        return A

    def visit(self, p: Axis):
        A = self._score(p)
        new = self.ops.solve_lsa_batched([A], maximize=True)[0] if self.device_lap else self.lsa_solver(A)
        pick = new.to(A.device)
        before, after = A.diag().sum(), A[torch.arange(A.shape[0], device=A.device), pick].sum()     # reference :80
        for sb in self.state_bs:
            apply_perm({p: pick}, self.spec, sb, inplace=True)
        # what the caller reads once per sweep: gain, "improved" by the reference's fp32 test (:81), and the matrix's norm
        report = torch.stack([after - before, (after > before + 1e-12).to(A.dtype), A.norm()])
        # the matrix goes to the caller (costs[p]); the grouped path reuses its buffer at the group's next visit
        return (A.clone() if self.grouped else A), new, report


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
            for p in order:
                costs[p], new, report = walker.visit(p)
                composed[p] = composed[p][new.to(composed[p].device)]
                log.append(report)
            rows = torch.stack(log).cpu()            # ONE read per sweep
            assert bool((rows[:, 2] > 0).all()), "weight_matching: a group's score matrix is zero"      # reference :77
            if verbose:
                for p, gain in zip(order, rows[:, 0].tolist()):
                    print("%d/%s:%d: %s" % (sweep, p.key, p.axis, gain))
            if not bool((rows[:, 1] > 0).any()):
                break
    perm = {p: v.cpu() for p, v in composed.items()}
    return (perm, costs) if return_costs else perm
