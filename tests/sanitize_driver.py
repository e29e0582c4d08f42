#!/usr/bin/env python3
"""Exercises the HOST code of libpleas_hip (plan builders, plan caches, XCD item ordering, lane dealing, the host LAP,
argument checks) under AddressSanitizer + UndefinedBehaviorSanitizer, WITHOUT a GPU.

Run by tests/test_host_logic.py::test_host_code_under_sanitizers as

    LD_PRELOAD=<libclang_rt.asan> ASAN_OPTIONS=detect_leaks=0 python tests/sanitize_driver.py <libpleas_hip_asan.so>

No torch, no numpy in this process (the preloaded runtime would instrument them too): ctypes and the C-ABI only.  Every
entry point called here returns before anything touches a device: the `*_ws_bytes` / `*_plan_info` functions build the
full launch plan of a layer / node list on the host, `pleas_lsap_host` is the host solver, and the compute entry points
are called with arguments their validation must refuse.  Prints SANITIZE_OK when every call returned what it should."""
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pleas_merging_amd import _lib  # noqa: E402  (ctypes only)

_lib.LIB_PATH = sys.argv[1]
lib = _lib.lib()


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sanitize_driver_layers import resnet_layers  # noqa: E402


def expect(cond, what):
    if not cond:
        print("SANITIZE_DRIVER_FAILED:", what, "| last error:", lib.pleas_last_error())
        sys.exit(1)


def plan_lists(layers, N, widen=1.0):
    n = len(layers)
    fwd, wg, neq = (_lib.FwdLayer * n)(), (_lib.WgradLayer * n)(), (_lib.NeqLayer * n)()
    kept = 0
    for (co, ci, h, w, k, s, p) in layers:
        co, ci = (int(co * widen), int(ci * widen) if ci > 3 else ci)
        f = fwd[kept]
        f.N, f.Cout, f.Cin, f.Hin, f.Win, f.KH, f.KW, f.stride, f.pad = N, co, ci, h, w, k, k, s, p
        f.Csrc, f.n_merged, f.flags = co, co, (1 if (k > 1 and ci % 32 == 0) else 0)
        g = wg[kept]
        g.N, g.Cout, g.Cin, g.Hin, g.Win, g.KH, g.KW, g.stride, g.pad = N, co, ci, h, w, k, k, s, p
        g.flags = 2 if (k > 1 and ci % 32 == 0) else 0
        q = neq[kept]
        q.N, q.Cin, q.Hin, q.Win, q.KH, q.KW, q.stride, q.pad = N, ci, h, w, k, k, s, p
        kept += 1
    return fwd, wg, neq, kept


def main():
    random.seed(0)
    expect(b"gfx950" in lib.pleas_version(), "version string")
    # ---- single-node workspace sizes, incl. degenerate shapes
    for B, C, HW in [(16, 256, 196), (16, 64, 12544), (1, 2048, 1), (3, 1000, 1), (16, 2048, 49), (2, 7, 5), (0, 256, 196),
                     (16, 0, 196), (16, 256, 0), (-1, 3, 3)]:
        got = lib.pleas_gram_ws_bytes(B, C, HW)
        expect((got > 0) == (B > 0 and C > 0 and HW > 0), "gram_ws_bytes(%d, %d, %d) = %d" % (B, C, HW, got))
    for n, c in [(16 * 196, 256), (2, 1), (0, 4), (4, 0), (1 << 30, 2048)]:
        lib.pleas_bn_train_ws_bytes(n, c)
    for n in (0, 1, 255, 256, 1 << 20, (1 << 31) + 5):
        lib.pleas_sqerr_ws_bytes(n)
    # ---- grouped launches: full host plans of the three architectures at several batch sizes and widths
    info = (ctypes.c_double * 4)()
    for arch in ("resnet18", "resnet50", "resnet101"):
        layers = resnet_layers(arch)
        expect(len(layers) == {"resnet18": 21, "resnet50": 54, "resnet101": 105}[arch], "layer count of " + arch)
        for N, widen in ((16, 1.0), (2, 1.0), (1, 1.0), (4, 1.5), (16, 2.0 - 1.0 / 64)):
            fwd, wg, neq, n = plan_lists(layers, N, widen)
            expect(lib.pleas_fwd_batch_ws_bytes(fwd, n) > 0, "fwd plan %s N=%d x%.2f" % (arch, N, widen))
            units = (ctypes.c_int * 96)()
            for ms in (None, (ctypes.c_double * 10)(1.8, 0.1, 0.1, 0.2, 2.6, 0.3, 1.6, 2.6, 0.2, 0.4), (ctypes.c_double * 10)(*([0.0] * 7 + [5.0, 0.0, 0.0]))):
                expect(0 < lib.pleas_fwd_plan_units(fwd, n, ms, units, 24) <= 24, "fwd launch units " + arch)
            # the weight gradient takes every layer, the 3-channel stem included (virtual-channel rows, csrc/conv.hip); the
            # normal-equation launch takes layers with >= 16 input channels
            expect(lib.pleas_wgrad_batch_ws_bytes(wg, n) > 0, "wgrad plan with the stem " + arch)
            expect(lib.pleas_wgrad_batch_ws_bytes(ctypes.byref(wg, ctypes.sizeof(_lib.WgradLayer)), n - 1) > 0, "wgrad plan " + arch)
            expect(lib.pleas_normal_eq_ws_bytes(ctypes.byref(neq, ctypes.sizeof(_lib.NeqLayer)), n - 1) > 0, "neq plan " + arch)
            expect(lib.pleas_normal_eq_plan_info(ctypes.byref(neq, ctypes.sizeof(_lib.NeqLayer)), n - 1, info) == 0 and info[1] > 0,
                   "neq plan info " + arch)
        # matching: one node per convolution output (+ its BatchNorm as a derived node), groups of the node's width
        nodes = (_lib.GramNode * (2 * len(layers)))()
        group_c = (ctypes.c_int * len(layers))()
        k = 0
        for g, (co, ci, h, w, ks, s, p) in enumerate(layers):
            ho = (h + 2 * p - ks) // s + 1
            group_c[g] = co
            nodes[k].B, nodes[k].C, nodes[k].HW, nodes[k].group = 16, co, ho * ho, g
            nodes[k + 1].B, nodes[k + 1].C, nodes[k + 1].HW, nodes[k + 1].group = 16, co, ho * ho, g
            nodes[k + 1].derived, nodes[k + 1].source = 1, k
            nodes[k + 1].scale_x = nodes[k + 1].shift_x = nodes[k + 1].scale_y = nodes[k + 1].shift_y = 16   # never read on the host
            k += 2
        expect(lib.pleas_gram_batch_ws_bytes(nodes, k, group_c, len(layers)) > 0, "gram batch plan " + arch)
        nodes[1].source = 7                      # a derived node whose source has another shape: refused, not followed
        expect(lib.pleas_gram_batch_ws_bytes(nodes, k, group_c, len(layers)) == 0, "gram batch plan must refuse a bad source")
        items = (_lib.MergeItem * len(layers))()
        for i, (co, ci, h, w, ks, s, p) in enumerate(layers):
            items[i].outer, items[i].inner, items[i].rows_out, items[i].rows_src, items[i].n_merged = 16, h * w, ci, ci, ci // 2
        expect(lib.pleas_merge_batch_ws_bytes(items, len(layers)) > 0, "merge batch plan " + arch)
    # ---- degenerate / hostile layer lists: must be refused (0 bytes / error code), never read out of bounds
    bad = (_lib.NeqLayer * 1)()
    for geo in [(0, 64, 14, 14, 3, 3, 1, 1), (4, 64, 2, 2, 7, 7, 1, 0), (4, 64, 14, 14, 3, 3, 0, 1), (4, -5, 14, 14, 1, 1, 1, 0),
                (1 << 20, 64, 1 << 10, 1 << 10, 1, 1, 1, 0)]:
        bad[0].N, bad[0].Cin, bad[0].Hin, bad[0].Win, bad[0].KH, bad[0].KW, bad[0].stride, bad[0].pad = geo
        expect(lib.pleas_normal_eq_ws_bytes(bad, 1) == 0, "neq plan must refuse %r" % (geo,))
        expect(lib.pleas_normal_eq_plan_info(bad, 1, info) != 0, "neq plan info must refuse %r" % (geo,))
    for geo in [(4, 64, 5, 6, 3, 3, 1, 1), (2, 20, 2, 3, 3, 3, 1, 1), (5, 24, 8, 8, 5, 5, 1, 2), (1, 16, 1, 1, 3, 3, 1, 1)]:
        bad[0].N, bad[0].Cin, bad[0].Hin, bad[0].Win, bad[0].KH, bad[0].KW, bad[0].stride, bad[0].pad = geo
        expect(lib.pleas_normal_eq_plan_info(bad, 1, info) == 0 and info[3] > 0, "lag classes of %r" % (geo,))
    expect(lib.pleas_normal_eq_finalize(None, 0, None) == -22, "finalize(NULL)")
    expect(lib.pleas_fwd_batch_ws_bytes(None, 0) == 0 and lib.pleas_wgrad_batch_ws_bytes(None, 3) == 0, "NULL layer lists")
    # ---- compute entry points with arguments their validation refuses (nothing is launched)
    expect(lib.pleas_gram_accum(None, None, 1, 4, 4, 0, 0, None, None, 0, None) == -22, "gram_accum(NULL)")
    n1 = (ctypes.c_int * 1)(4097)
    ptr = (ctypes.c_void_p * 1)(8)
    expect(lib.pleas_lsap_batched(ptr, n1, 1, 1, ptr, None) == -22, "lsap_batched(n > max)")
    expect(lib.pleas_bn_act_maxpool(ptr, None, None, ptr, 2, 4, 8, 8, 3, 3, 2, 2, 1, None) == -22, "bn_act_maxpool(pad > kernel / 2)")
    expect(lib.pleas_bn_act_maxpool(ptr, ptr, None, ptr, 2, 4, 8, 8, 3, 3, 2, 1, 1, None) == -22, "bn_act_maxpool(scale without shift)")
    expect(lib.pleas_bn_act_maxpool(ptr, None, None, ptr, 2, 4, 2, 2, 7, 7, 1, 1, 1, None) == -22, "bn_act_maxpool(window > input)")
    expect(lib.pleas_bn_train_fold_batches(None, 4, 2, 8, 16, None, None, 1e-5, 0.1, None, None, None, None, None, None, 0, None) == -22,
           "bn_train_fold_batches(NULL)")
    expect(lib.pleas_bn_act_tracked_batches(ptr, ptr, ptr, None, None, None, ptr, 4, 0, 8, 16, 1, None) == -22, "bn_act_tracked_batches(0 batches)")
    expect(lib.pleas_masked_adam(None, None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 1, None) == -22, "masked_adam(NULL)")
    # ---- host LAP: random, tie-heavy, all-equal, n = 1; fp32 and fp64; result must be a permutation of maximal value
    for n in (1, 2, 3, 17, 64, 129):
        for kind in ("normal", "ties", "equal"):
            vals = [random.gauss(0, 1) if kind == "normal" else float(random.randint(0, 2)) if kind == "ties" else 0.5
                    for _ in range(n * n)]
            for is_double, ctype in ((0, ctypes.c_float), (1, ctypes.c_double)):
                cost = (ctype * (n * n))(*vals)
                col = (ctypes.c_int64 * n)()
                for maximize in (1, 0):
                    expect(lib.pleas_lsap_host(cost, is_double, n, maximize, col) == 0, "lsap_host n=%d" % n)
                    expect(sorted(col) == list(range(n)), "lsap_host result is a permutation (n=%d, %s)" % (n, kind))
    expect(lib.pleas_lsap_host(None, 0, 4, 1, None) == -22, "lsap_host(NULL)")
    print("SANITIZE_OK")


if __name__ == "__main__":
    main()
