from pleas_merging_amd.methods.activation_matching import (  # noqa: F401
    activation_matching, build_cross_module, compute_matching_costs, cross_features_cdist,
    cross_features_inner_product, shard_batches, allreduce_sum_)
