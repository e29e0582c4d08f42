from pleas_merging_amd.methods import *  # noqa: F401,F403
from pleas_merging_amd.methods import (activation_matching, cross_features_cdist, cross_features_inner_product,  # noqa: F401
                                       weight_matching, partial_merge, get_blocks, expand_ratios, train)
