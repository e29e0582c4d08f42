"""Drop-in for the reference's ``pleas/methods/__init__.py`` (:12-41): the same names, bound the same way -- each
``from pleas.methods.<module> import <function>`` first imports the submodule and then rebinds the package attribute
to the FUNCTION, so ``pleas.methods.activation_matching`` is the function here as it is there."""
from pleas_merging_amd.methods import *  # noqa: F401,F403  (additions: reset_bn_stats, zip_ratios, ...)
from pleas.methods.activation_matching import (  # noqa: F401
    activation_matching, cross_features_cdist, cross_features_inner_product)
from pleas.methods.weight_matching import weight_matching  # noqa: F401
from pleas.methods.partial_matching import (  # noqa: F401
    partial_merge, get_blocks, qp_ratios, expand_ratios, partial_merge_flops)
from pleas.methods.pleas_merging import (  # noqa: F401
    train, train_eval_linear_probe, eval_perm_model, eval_whole_model, get_fc_perm)
import pleas.methods.extras  # noqa: F401,E402
