"""Linear-assignment solvers with the reference's call signature ``solver(A) -> LongTensor[n]``.

``hip_solve_lsa`` (default of this package) runs the batched gfx950 kernel;
``scipy_solve_lsa`` keeps the reference's public symbol (pleas/core/solvers.py:18-33,
re-exported at pleas/core/__init__.py:21) for callers that pass it explicitly.
"""
from __future__ import annotations

import torch


def hip_solve_lsa(A: torch.Tensor, maximize: bool = True) -> torch.Tensor:
    from ..hip_ops import hip_solve_lsa as _impl

    return _impl(A, maximize)


def host_solve_lsa(A: torch.Tensor, maximize: bool = True) -> torch.Tensor:
    """The library's own solver on a HOST cost matrix (``pleas_lsap_host``, scipy's scan order and tie rule): what a
    caller with CPU state dicts passes as ``lsa_solver`` -- explicitly; device tensors always take ``hip_solve_lsa``."""
    from ..hip_ops import host_solve_lsa as _impl

    return _impl(A, maximize)


def scipy_solve_lsa(A: torch.Tensor, maximize: bool = True) -> torch.Tensor:
    """Host solver of the reference (pleas/core/solvers.py:18-33); never a default here."""
    import scipy.optimize

    ri, ci = scipy.optimize.linear_sum_assignment(A.detach().cpu().numpy(), maximize=maximize)
    ri, ci = torch.as_tensor(ri), torch.as_tensor(ci)
    assert (ri == torch.arange(len(ri))).all()
    return ci
