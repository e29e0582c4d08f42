from pleas_merging_amd.methods.partial_matching import (  # noqa: F401
    expand_ratios, get_blocks, build_partial_merge_model, partial_merge)
