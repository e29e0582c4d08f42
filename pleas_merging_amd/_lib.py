"""ctypes binding of libpleas_hip.so (the C-ABI in include/pleas_hip.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p
from typing import Optional

_LIB: Optional[ctypes.CDLL] = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpleas_hip.so")

EPI_INNER, EPI_NEG_CDIST = 0, 1
LSAP_MAX_N = 4096

# name -> (restype, argtypes); one row per symbol declared in include/pleas_hip.h
SIGNATURES = {
    "pleas_version": (c_char_p, []),
    "pleas_last_error": (c_char_p, []),
    "pleas_gram_ws_bytes": (c_size_t, [c_int, c_int, c_int64]),
    "pleas_gram_accum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "pleas_lsap_batched": (c_int, [POINTER(c_void_p), POINTER(c_int), c_int, c_int, POINTER(c_void_p), c_void_p]),
    "pleas_lsap_host": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "pleas_merge_blocks": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pleas_merge_batch_ws_bytes": (c_size_t, [c_void_p, c_int]),
    "pleas_merge_batch": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "pleas_bn_act": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p]),
    "pleas_bn_act_tracked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                     c_int64, c_int, c_void_p]),
    "pleas_bn_act_maxpool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_void_p]),
    "pleas_bn_act_tracked_batches": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                             c_int, c_int64, c_int, c_void_p]),
    "pleas_bn_train_fold_batches": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, ctypes.c_double,
                                            ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_size_t, c_void_p]),
    "pleas_bn_train_ws_bytes": (c_size_t, [c_int64, c_int]),
    "pleas_bn_train_fold": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pleas_masked_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                  c_float, c_int, c_void_p]),
    "pleas_sqerr_ws_bytes": (c_size_t, [c_int64]),
    "pleas_sqerr": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_int, c_void_p, c_float, c_void_p, c_void_p, c_size_t,
                            c_void_p]),
    "pleas_channel_sum": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "pleas_prof_enable": (None, [c_int]),
    "pleas_prof_select": (None, [ctypes.c_uint]),
    "pleas_prof_reset": (None, []),
    "pleas_prof_collect": (c_int, [c_int, POINTER(c_int64), POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                   POINTER(ctypes.c_double)]),
    "pleas_gram_tune": (None, [c_int, c_int]),
    "pleas_gram_split_bf16": (None, [c_int]),
    "pleas_arith": (None, [c_int]),
    "pleas_arith_get": (c_int, []),
    "pleas_gram_batch_tune": (None, [c_int, c_int]),
    "pleas_wgrad_tune": (None, [c_int]),
    "pleas_target_residual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int64, c_float, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "pleas_target_residual_max_partials": (c_int, []),
    "pleas_loss_final": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pleas_cholesky_solve_batched": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int), POINTER(c_int), c_int,
                                             c_float, c_void_p, c_void_p]),
    "pleas_normal_eq_ws_bytes": (c_size_t, [c_void_p, c_int]),
    "pleas_normal_eq_accum": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "pleas_normal_eq_finalize": (c_int, [c_void_p, c_int, c_void_p]),
    "pleas_normal_eq_plan_info": (c_int, [c_void_p, c_int, c_void_p]),
    "pleas_fwd_batch_ws_bytes": (c_size_t, [c_void_p, c_int]),
    "pleas_fwd_plan_lanes": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pleas_fwd_plan_units": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int]),
    "pleas_allreduce_sum": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "pleas_fwd_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "pleas_conv2d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "pleas_conv2d_bn_act_fwd": (c_int, [c_void_p] * 8 + [c_int] * 11 + [c_void_p]),
    "pleas_wgrad_batch_ws_bytes": (c_size_t, [c_void_p, c_int]),
    "pleas_wgrad_batch": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "pleas_gram_batch_ws_bytes": (c_size_t, [c_void_p, c_int, POINTER(c_int), c_int]),
    "pleas_gram_batch": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int), c_int, c_int, c_int, c_void_p,
                                 c_size_t, c_int, c_void_p]),
}


class MergeItem(ctypes.Structure):
    """struct pleas_merge_item"""
    _fields_ = [("w1", c_void_p), ("w2", c_void_p), ("out", c_void_p), ("row1", c_void_p), ("row2", c_void_p),
                ("outer", c_int64), ("inner", c_int64), ("rows_out", c_int), ("rows_src", c_int), ("n_merged", c_int),
                ("sub_stride", c_int), ("sub_h", c_int), ("sub_w", c_int)]


class FwdLayer(ctypes.Structure):
    """struct pleas_fwd_layer"""
    _fields_ = [("ip", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("o1", c_void_p), ("o2", c_void_p),
                ("row1", c_void_p), ("row2", c_void_p), ("resid", c_void_p), ("N", c_int), ("Cout", c_int), ("Cin", c_int),
                ("Hin", c_int), ("Win", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad", c_int),
                ("Csrc", c_int), ("n_merged", c_int), ("dscale", c_float), ("loss_scale", c_float), ("flags", c_int)]


class NeqLayer(ctypes.Structure):
    """struct pleas_neq_layer"""
    _fields_ = [("ip", c_void_p), ("A", c_void_p), ("N", c_int), ("Cin", c_int), ("Hin", c_int), ("Win", c_int),
                ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad", c_int)]


class WgradLayer(ctypes.Structure):
    """struct pleas_wgrad_layer"""
    _fields_ = [("resid", c_void_p), ("ip", c_void_p), ("grad", c_void_p), ("N", c_int), ("Cout", c_int), ("Cin", c_int),
                ("Hin", c_int), ("Win", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad", c_int),
                ("flags", c_int)]


class GramNode(ctypes.Structure):
    """struct pleas_gram_node"""
    _fields_ = [("x", c_void_p), ("y", c_void_p), ("B", c_int), ("C", c_int), ("HW", c_int64), ("group", c_int),
                ("derived", c_int), ("source", c_int), ("scale_x", c_void_p), ("shift_x", c_void_p),
                ("scale_y", c_void_p), ("shift_y", c_void_p)]

PROF_KERNELS = ["gram_partial", "gram_finalize", "lsap", "merge_blocks", "masked_adam", "sqerr", "conv_fwd",
                "conv_wgrad", "normal_eq", "solve", "bn_act", "conv2d"]


class PleasHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load the library once; raise loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PleasHipError(
                "libpleas_hip.so is missing (%s). Build it with `python -m pleas_merging_amd.build` "
                "(needs hipcc); there is no CPU fallback for the HIP path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the build lacks a declared symbol
            fn.restype, fn.argtypes = restype, argtypes
        _LIB = handle
    return _LIB


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().pleas_last_error().decode() or "error code %d" % rc
        raise PleasHipError("%s failed (%d): %s" % (what, rc, msg))
