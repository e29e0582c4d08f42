"""Core types of the matching/merging path (mirrors reference pleas/core/__init__.py:9-21)."""
from .utils import (
    Axis,
    PermutationGroup,
    PermutationSpec,
    Permutation,
    apply_perm,
    make_identity_perm,
    make_random_perm,
    invert_perm,
    perm_eq,
)
from .compiler import get_permutation_spec
from .solvers import hip_solve_lsa, host_solve_lsa, scipy_solve_lsa
