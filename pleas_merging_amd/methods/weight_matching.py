"""Weight matching (Git Re-Basin coordinate descent) on the HIP kernels.

Drop-in for the reference's ``pleas/methods/weight_matching.py:22-95``: sweep the groups in
a seeded random order; for each group sum ``Wa . Wb^T`` over its state axes, solve the LAP,
permute model B's tensors, repeat until no group improves.

HIP path: the per-axis inner products accumulate directly into one n x n matrix with
``pleas_gram_accum`` (``accumulate=1``; any tensor axis maps onto its [B][C][HW] view, so no
movedim/reshape copies), and each LAP is one ``pleas_lsap_batched`` launch.  The sweep itself
is inherently sequential (every LAP changes model B before the next group is scored).
Other ``cross_weights`` / ``lsa_solver`` callables go through the reference's plug points.
"""
from __future__ import annotations

from copy import copy, deepcopy
from typing import Dict, Sequence, Union

import torch

from ..core.solvers import hip_solve_lsa
from ..core.utils import Permutation, PermutationSpec, StateDict, apply_perm, make_identity_perm
from ..hip_ops import cross_features_inner_product


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
    """Same signature, defaults and return value as the reference (``perm`` or ``(perm, costs)``;
    ``costs[p]`` is the last score matrix of group ``p``, in already-permuted-B coordinates)."""
    if isinstance(state_as, dict):
        state_as = [state_as]
    if isinstance(state_bs, dict):
        state_bs = [state_bs]
    assert len(state_as) == len(state_bs)
    if not inplace:
        state_bs = [copy(sb) for sb in state_bs]

    perm = make_identity_perm(spec) if init_perm is None else deepcopy(init_perm)
    if init_perm is not None:
        for sb in state_bs:
            apply_perm(init_perm, spec, sb, inplace=True)

    names = list(perm.keys())
    device = next(iter(state_as[0].values())).device
    fused = cross_weights is cross_features_inner_product
    if fused:
        from .. import hip_ops

        if device.type != "cuda":
            raise hip_ops.PleasHipError(
                "weight_matching: state dicts are on %s; the default HIP cross_weights needs GPU tensors "
                "(pass explicit cross_weights= / lsa_solver= callables to run elsewhere)" % device)
    skip = tuple(skip_suffixes)
    costs: Dict = {}
    rng = torch.Generator()
    rng.manual_seed(seed)

    with torch.no_grad():
        for sweep in range(max_iter):
            progress = False
            for ix in torch.randperm(len(names), generator=rng):
                p = names[ix]
                group = spec[p]
                n = group.size
                A = torch.zeros(n, n, device=device)
                for ax in group.state:
                    if ax.key.endswith(skip):
                        continue
                    for sa, sb in zip(state_as, state_bs):
                        if skip_missing and not (ax.key in sa and ax.key in sb):
                            continue
                        if fused:
                            hip_ops.gram_accum(sa[ax.key], sb[ax.key], ax.axis, A, hip_ops.EPI_INNER, True)
                        else:
                            A.add_(cross_weights(sa[ax.key], sb[ax.key], ax.axis))
                assert A.norm() > 0
                new = lsa_solver(A)
                idx = torch.arange(n, device=A.device)
                old_l, new_l = A.diag().sum(), A[idx, new.to(A.device)].sum()
                progress = progress or bool(new_l > old_l + 1e-12)
                if verbose:
                    print("%d/%s:%d: %s" % (sweep, p.key, p.axis, float(new_l - old_l)))
                perm[p] = perm[p][new]
                costs[p] = A
                for sb in state_bs:
                    apply_perm({p: new}, spec, sb, inplace=True)
            if not progress:
                break
    return (perm, costs) if return_costs else perm
