from pleas_merging_amd.methods.pleas_merging import (  # noqa: F401
    get_gradient_mask, train, ActivationTap, PleasFitter, cosine_lrs, dp_slice, dp_sum_)
