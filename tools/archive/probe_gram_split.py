"""STUDY: can the fp32-MFMA roof of the matching contraction be moved by bf16 MFMA on a three-way
bf16 split of the fp32 operands (six products per fp32 product, fp32 accumulate)?  Off by default in the library
(`pleas_gram_split_bf16`); this script measures, against an fp64 contraction of the SAME fp32 operands:

  1. the 13 GRAM_SHAPES of tests/test_hip_kernels.py: error of the inner product and of the distance epilogue, exact
     fp32-MFMA path vs split path, and the time of each;
  2. ResNet-101 matching costs: the tracked activations of `batches` batches, every node contracted three ways (exact,
     split, fp64) from the same tensors (the generic `build_cross_module` path, so that the vendor convolutions' run-to-run
     differences do not enter), summed per group: error vs fp64 per group, and the LAP assignments of the three;
  3. the grouped launch of the product path (accumulate_costs_fused) with the switch off / on: time per batch.

Usage: python tools/probe_gram_split.py [--batches 8] [--arch resnet101] [--out gpurun_out/r03_gram_split.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from pleas_merging_amd import _lib, hip_ops
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods import activation_matching as am_pkg  # noqa: F401  (the function; module below)
import importlib

am = importlib.import_module("pleas_merging_amd.methods.activation_matching")

GRAM_SHAPES = [((4, 8, 6, 6), 1), ((3, 20, 7, 7), 1), ((2, 64, 28, 28), 1), ((2, 96, 14, 14), 1), ((16, 256, 14, 14), 1),
               ((5, 130, 9, 9), 1), ((16, 512, 1, 1), 1), ((16, 300), 1), ((64, 32, 3, 3), 0), ((64, 32, 3, 3), 1),
               ((10, 48), 0), ((48,), 0), ((2, 64, 112, 112), 1)]
EXTRA_SHAPES = [((16, 1024, 14, 14), 1), ((16, 512, 28, 28), 1), ((16, 2048, 7, 7), 1), ((16, 256, 56, 56), 1),
                ((64, 1024, 14, 14), 1), ((64, 2048, 7, 7), 1)]


def split(on):
    _lib.lib().pleas_gram_split_bf16(int(on))


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def fp64_pair(x, y, axis):
    """(inner products, negative distances) of the fp32 operands, evaluated in fp64 on the device."""
    xd = x.double().movedim(axis, 0).reshape(x.shape[axis], -1) if x.dim() > 1 else x.double().reshape(-1, 1)
    yd = y.double().movedim(axis, 0).reshape(y.shape[axis], -1) if y.dim() > 1 else y.double().reshape(-1, 1)
    return xd @ yd.T


def fp64_cdist_batchwise(x, y, axis):
    """Negative Euclidean distance between feature rows, in fp64 (what the distance epilogue approximates)."""
    xd = x.double().movedim(axis, 0).reshape(x.shape[axis], -1) if x.dim() > 1 else x.double().reshape(-1, 1)
    yd = y.double().movedim(axis, 0).reshape(y.shape[axis], -1) if y.dim() > 1 else y.double().reshape(-1, 1)
    g = xd @ yd.T
    d2 = (xd * xd).sum(1)[:, None] + (yd * yd).sum(1)[None, :] - 2 * g
    return -d2.clamp_min(0).sqrt()


def timed(fn, reps=20):
    """Microseconds per launch of the contraction kernel alone (the library's event pairs around `gram_partial`)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    hip_ops.profile_enable(True)
    hip_ops.profile_reset()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    launches, ms, _, _ = hip_ops.profile_collect()["gram_partial"]
    hip_ops.profile_enable(False)
    return 1e-3 * ms / launches


def study_shapes():
    rows = []
    for shape, axis in GRAM_SHAPES + EXTRA_SHAPES:
        g = torch.Generator().manual_seed(hash(shape) % 1000)
        x = torch.randn(shape, generator=g)
        y = 0.7 * x + 0.5 * torch.randn(shape, generator=g)
        x, y = x.cuda(), y.cuda()
        want_g, want_d = fp64_pair(x, y, axis), fp64_cdist_batchwise(x, y, axis)
        row = {"shape": list(shape), "axis": axis}
        for name, on in (("exact", 0), ("split", 1)):
            split(on)
            got_g = hip_ops.cross_features_inner_product(x, y, axis)
            got_d = hip_ops.cross_features_cdist(x, y, axis)
            row[name] = {"inner_rel_fro": rel(got_g, want_g), "inner_max_abs": float((got_g.double() - want_g).abs().max()),
                         "cdist_rel_fro": rel(got_d, want_d),
                         "kernel_us": 1e6 * timed(lambda: hip_ops.cross_features_inner_product(x, y, axis))}
        split(0)
        rows.append(row)
        print("%-20s axis %d | inner rel-fro exact %.2e split %.2e | cdist exact %.2e split %.2e | %7.1f us -> %7.1f us" % (
            shape, axis, row["exact"]["inner_rel_fro"], row["split"]["inner_rel_fro"], row["exact"]["cdist_rel_fro"],
            row["split"]["cdist_rel_fro"], row["exact"]["kernel_us"], row["split"]["kernel_us"]), flush=True)
    return rows


def make_pair(arch, device):
    import bench      # the bench's pair: random init, BatchNorm statistics calibrated

    return bench.build_models(arch, device, 16)


def study_costs(arch, n_batches):
    dev = torch.device("cuda")
    m1, m2 = make_pair(arch, dev)
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    g = torch.Generator().manual_seed(11)
    data = [(torch.randn(16, 3, 224, 224, generator=g), None) for _ in range(n_batches)]

    def three_ways(x, y, axis):
        split(0)
        a = hip_ops.cross_features_cdist(x, y, axis).double()
        split(1)
        b = hip_ops.cross_features_cdist(x, y, axis).double()
        split(0)
        return torch.stack([a, b, fp64_cdist_batchwise(x, y, axis)])

    axes = [ax for group in spec.values() for ax in group.node]
    gm = am.build_cross_module(m1, m2, axes, three_ways)
    costs = am.compute_matching_costs(spec, gm, data, n_batches, True, dev, shard=False)
    groups = []
    keys = list(costs.keys())
    perms = {}
    for which, name in enumerate(("exact", "split", "fp64")):
        mats = [costs[k][which].float().contiguous() if name != "fp64" else costs[k][which].contiguous() for k in keys]
        if name == "fp64":      # the LAP kernel takes fp32; solve the fp64 costs on the host
            from scipy.optimize import linear_sum_assignment
            perms[name] = [torch.as_tensor(linear_sum_assignment(m.cpu().numpy(), maximize=True)[1]) for m in mats]
        else:
            perms[name] = [p.cpu() for p in hip_ops.solve_lsa_batched(mats, maximize=True)]
    worst = {"exact": 0.0, "split": 0.0}
    for i, k in enumerate(keys):
        e, s = rel(costs[k][0], costs[k][2]), rel(costs[k][1], costs[k][2])
        worst["exact"], worst["split"] = max(worst["exact"], e), max(worst["split"], s)
        groups.append({"group": "%s:%d" % (k.key, k.axis), "size": int(costs[k].shape[-1]), "exact_rel_fro": e, "split_rel_fro": s,
                       "split_vs_exact_rel_fro": rel(costs[k][1], costs[k][0]),
                       "same_assignment_split_exact": bool(torch.equal(perms["exact"][i], perms["split"][i])),
                       "same_assignment_exact_fp64": bool(torch.equal(perms["exact"][i], perms["fp64"][i])),
                       "same_assignment_split_fp64": bool(torch.equal(perms["split"][i], perms["fp64"][i]))})
    n = len(groups)
    summary = {"arch": arch, "batches": n_batches, "groups": n,
               "worst_rel_fro_vs_fp64": worst,
               "median_rel_fro_vs_fp64": {w: sorted(gr[w + "_rel_fro"] for gr in groups)[n // 2] for w in ("exact", "split")},
               "groups_split_error_above_exact": sum(1 for gr in groups if gr["split_rel_fro"] > gr["exact_rel_fro"]),
               "assignments_identical_split_vs_exact": sum(gr["same_assignment_split_exact"] for gr in groups),
               "assignments_identical_exact_vs_fp64": sum(gr["same_assignment_exact_fp64"] for gr in groups),
               "assignments_identical_split_vs_fp64": sum(gr["same_assignment_split_fp64"] for gr in groups)}
    print(json.dumps(summary), flush=True)
    # ---- the product path's grouped launch, switch off / on: time per batch of the whole matching pass
    from pleas_merging_amd.hip_ops import EPI_NEG_CDIST
    timing = {}
    for name, on in (("exact", 0), ("split", 1), ("exact_again", 0)):
        split(on)
        am.accumulate_costs_fused(spec, m1, m2, data, 4, EPI_NEG_CDIST, batches_per_forward=4)      # plans, warm-up
        torch.cuda.synchronize()
        hip_ops.profile_enable(True)
        hip_ops.profile_reset()
        t0 = time.perf_counter()
        am.accumulate_costs_fused(spec, m1, m2, data, n_batches, EPI_NEG_CDIST, batches_per_forward=4)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        launches, ms, flops, _ = hip_ops.profile_collect()["gram_partial"]
        hip_ops.profile_enable(False)
        timing[name] = {"ms_per_batch": 1e3 * wall / n_batches, "gram_launches": launches, "gram_ms_per_launch": ms / launches,
                        "gram_algorithmic_tflops": flops / (ms * 1e-3) / 1e12,
                        "fraction_of_fp32_mfma_peak_157.3": flops / (ms * 1e-3) / 1e12 / 157.3}
    split(0)
    print(json.dumps(timing), flush=True)
    return {"summary": summary, "groups": groups, "matching_pass": timing}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    out = {"shapes": study_shapes(), "costs": study_costs(args.arch, args.batches)}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
