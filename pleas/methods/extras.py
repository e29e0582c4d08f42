from pleas_merging_amd.methods.extras import (  # noqa: F401
    reset_bn_stats, zip_ratios, save_matching, load_matching, load_checkpoint)
