from pleas_merging_amd.methods.pleas_merging import (  # noqa: F401
    get_gradient_mask, train, ActivationTap, FrozenSources, PleasFitter, prepare_sources, cosine_lrs, dp_slice, dp_sum_)
from pleas_merging_amd.methods.evaluation import (  # noqa: F401  (reference :408-496, :575-586)
    get_fc_perm, permute_final_features, eval_perm_model, eval_whole_model, train_eval_linear_probe)
