from pleas_merging_amd.core import *  # noqa: F401,F403
from pleas_merging_amd.core import (Axis, PermutationGroup, PermutationSpec, Permutation, apply_perm,  # noqa: F401
                                    make_identity_perm, make_random_perm, invert_perm, scipy_solve_lsa, hip_solve_lsa,
                                    host_solve_lsa)
from pleas.core.utils import count_linear_flops  # noqa: F401,E402  (reference pleas/core/__init__.py:9-19)
