"""MI355X-native PLeaS activation-matching + least-squares merging hot path.

Host side mirrors the reference's ``pleas.core`` / ``pleas.methods`` API; the
compute is done by hand-written gfx950 HIP kernels behind the C-ABI declared in
``include/pleas_hip.h`` (library ``pleas_merging_amd/csrc/libpleas_hip.so``).
"""
__version__ = "0.3.0"
