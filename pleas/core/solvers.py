from pleas_merging_amd.core.solvers import *  # noqa: F401,F403
from pleas_merging_amd.core.solvers import scipy_solve_lsa, hip_solve_lsa  # noqa: F401
