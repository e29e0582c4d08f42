"""Matching / merging methods (mirrors reference pleas/methods/__init__.py:12-35)."""
from .activation_matching import (
    activation_matching,
    build_cross_module,
    compute_matching_costs,
    cross_features_cdist,
    cross_features_inner_product,
)
from .weight_matching import weight_matching
from .partial_matching import expand_ratios, get_blocks, partial_merge, build_partial_merge_model
from .pleas_merging import train, get_gradient_mask
from .extras import reset_bn_stats, zip_ratios, save_matching, load_matching, load_checkpoint
from .budget import count_linear_flops, partial_merge_flops, qp_ratios
from .evaluation import (get_fc_perm, permute_final_features, eval_perm_model, eval_whole_model,
                         train_eval_linear_probe)
