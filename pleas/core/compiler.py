from pleas_merging_amd.core.compiler import *  # noqa: F401,F403
from pleas_merging_amd.core.compiler import get_permutation_spec, check_permutation_spec  # noqa: F401
