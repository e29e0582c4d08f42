from pleas_merging_amd.methods.weight_matching import weight_matching  # noqa: F401
