from pleas_merging_amd.methods.partial_matching import (  # noqa: F401
    expand_ratios, get_blocks, spread_blocks, block_maps, merged_state, build_partial_merge_model, partial_merge)
from pleas_merging_amd.methods.budget import partial_merge_flops, qp_ratios  # noqa: F401  (reference :205-257)
