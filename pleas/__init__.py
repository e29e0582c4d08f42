"""Drop-in namespace: ``import pleas`` resolves to the MI355X-native implementation.

A user of the reference keeps ``from pleas.core.compiler import get_permutation_spec`` /
``from pleas.methods.activation_matching import activation_matching`` ... unchanged; every
module here only re-exports ``pleas_merging_amd``.
"""
from pleas_merging_amd import __version__  # noqa: F401
