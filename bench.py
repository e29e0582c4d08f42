#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): wall-clock seconds to merge a ResNet-101 pair --
100-batch activation matching + LAP (71 groups) + partial merge + 400-step PLeaS (401 Adam updates) --
on synthetic 224x224 inputs, on N MI355X GPUs of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--arch resnet101] [--batch 16]

A STEP is ONE WHOLE JOB (100 matching batches + 71 LAPs + merge + 401 updates on one set of synthetic batches that is
resident in HBM).  ``--steps K`` times exactly K jobs back to back, bracketed by barrier + synchronize on both sides
(max over ranks); ``value`` = that time / K = seconds per job = BASELINE.json's metric; ``ms_per_step`` = 1000 * value.
``--warmup W`` runs W untimed jobs first.  ``--match-batches`` / ``--updates`` shorten the job for debugging only (the
line then says SHORTENED and is not a benchmark result).

N > 1: the driver starts one rank per GPU with ``python -m torch.distributed.run ... bench.py --gpus N``; when started
WITHOUT that launcher (``python bench.py --gpus N``) this script starts it itself, before anything touches the GPU, and
exits with its status.  Strong scaling, total work fixed: matching shards whole batches over ranks and all-reduces the
flat cost arena once (RCCL); each PLeaS update shards the batch's samples over ranks and all-reduces the flat gradient
arena, so every rank applies the identical update.

Besides the contract's keys the JSON line carries
  roofline      the own kernel with the most time in the timed jobs: algorithmic flop per launch / HIP-event time of
                its launches inside the timed region, against the 157.3 TFLOP/s fp32 matrix peak (+ roofline_other);
  cpu_baseline  the CPU oracle (restatement of the reference) on this box's host cores, on the job's own first batches
                at the job's batch size (bounded count), N = 1 only; the same sample gives checks.parity_vs_oracle;
  phases_s      one more (untimed, synchronised) job split into spec / matching / LAP / merge + set-up / updates;
  alt_solver    the closed-form PLeaS phase (solver="normal_eq": MFMA normal equations + batched Cholesky), N = 1;
  library_baseline  the reference's loops with STOCK PyTorch-ROCm operators on the same GPU (torch.cdist, autograd convolutions,
                torch.optim.Adam, scipy LAP after D2H), bounded sample scaled like cpu_baseline, N = 1 only;
  vendor        the frozen source forwards of the PLeaS phase (own convolutions since round 5 + pleas_bn_act) timed alone;
  checks        invariants of the last timed job's result (permutations valid, losses fell, weights finite) and
                parity_vs_oracle: the HIP job with the timed knobs against the oracle on the same batches (exit 3 on failure).
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
FULL_MATCH, FULL_PLEAS = 100, 401
T_START = time.time()


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed jobs (a step is one whole matching + PLeaS job)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed jobs before the timed ones")
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--ratio", type=float, default=0.0)
    ap.add_argument("--match-batches", type=int, default=FULL_MATCH, help="debug: matching batches per job (default 100)")
    ap.add_argument("--updates", type=int, default=FULL_PLEAS, help="debug: PLeaS updates per job (default 401)")
    ap.add_argument("--inputs", default="resident", choices=("resident", "host"),
                    help="where the timed jobs take their batches from: HBM-resident tensors, or pinned host tensors copied "
                         "host -> device inside the loops on a copy stream; the other mode is timed on two extra jobs and "
                         "reported beside it (`inputs`)")
    ap.add_argument("--match-mode", default="eval", choices=("eval", "train"),
                    help="train: both models are put in train mode before every job's activation matching, as the reference "
                         "drivers run it (no .eval() before activation_matching, run_domainnet.py:172-186): BatchNorm uses "
                         "batch statistics and moves its running statistics; `train` puts the sources in eval mode as ever")
    ap.add_argument("--solver", default="adam", choices=("adam", "normal_eq"),
                    help="adam (default, the headline): the reference's optimiser, 401 updates; normal_eq: the closed form of the "
                         "same objective as the PLeaS phase of every job (normal equations accumulated over this rank's whole "
                         "batches b %% N == rank, ONE all-reduce of the A / B arenas, layer-sharded batched Cholesky, ONE sum of "
                         "the solved parameter arena) -- the path whose exchange does not grow with the updates; the line is "
                         "tagged ALT-SOLVER and is not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-solver", action="store_true", help="skip the closed-form (normal equations) leg")
    ap.add_argument("--no-library-baseline", action="store_true", help="skip the stock-PyTorch-ROCm-operators leg on the GPU")
    ap.add_argument("--no-alt-arith", action="store_true", help="skip the split-bf16 leg (the same job under pleas_arith(1))")
    ap.add_argument("--no-phases", action="store_true", help="skip the extra synchronised job that fills phases_s")
    ap.add_argument("--phase-log", action="store_true", help="debug: log the phases of that job as they finish")
    ap.add_argument("--prefetch-groups", type=int, default=2,
                    help="groups of source forwards enqueued while the LAP kernel runs (0: none); also bounded by memory.  "
                         "Two 128-sample groups take about as long as the LAP kernel (0.16 s); more would still be running "
                         "beside the first updates and stretch their kernels (fwd 2.7 -> 3.2 ms at 6 groups, same job time)")
    ap.add_argument("--sources-per-forward", type=int, default=10,
                    help="updates PER RANK whose batches go through the frozen sources as ONE forward (160 samples per "
                         "forward at 10: a layer's grid is then a few rounds of the chip's 512 workgroup slots instead of a "
                         "fraction of one; 8 -> 10: -1.6 %% of the job, 12: no further gain for 28 GB more taps, "
                         "profiles/r05_exp_conv_bn.txt); 0 = the fitter's default of 2; the prefetch takes "
                         "--prefetch-groups forwards of that size")
    ap.add_argument("--match-per-forward", type=int, default=10,
                    help="matching batches per twin forward (every tracked node is still contracted per batch): the vendor "
                         "convolutions run faster per sample at 64-160 samples than at 16; 0 = the library's default (forwards of up to 64 "
                         "samples, at most 4 batches)")
    ap.add_argument("--miopen-find", type=int, default=0, help="1: torch.backends.cudnn.benchmark = True, i.e. the vendor "
                    "library times its candidate convolution kernels per configuration (Find mode) instead of taking the "
                    "immediate-mode pick; costs seconds per new configuration in the first warm-up job")
    ap.add_argument("--no-shard-optimizer", action="store_true", help="data parallel: all-reduce + full Adam on every rank "
                    "instead of the default (reduce-scatter the gradient arena, Adam on this rank's 1/world slice, all-gather "
                    "the parameters)")
    ap.add_argument("--prefetch-memory", type=float, default=0.7, help="share of the free HBM the prefetched taps may take")
    ap.add_argument("--emulate-allreduce-us", type=float, default=0.0,
                    help="with --emulate-world: hold the update stream this long where the gradient all-reduce would run")
    ap.add_argument("--lookahead", type=int, default=-1, help="1: the next group's source forwards are enqueued before the "
                    "current group's PLeaS updates and share the GPU with them; 0: one group at a time; -1 (default): 0 on "
                    "one GPU, 1 under data parallelism (fills the all-reduce gaps)")
    ap.add_argument("--profile-all", action="store_true", help="also bracket the many-launch elementwise kernel "
                    "(bn_act) with events: complete kernels_ms, slightly slower timed region")
    ap.add_argument("--cpu-match-batches", type=int, default=10, help="cpu_baseline / parity sample: matching batches of the "
                    "job's own (default 10 = one twin forward of the timed size)")
    ap.add_argument("--cpu-updates", type=int, default=8, help="cpu_baseline / parity sample: PLeaS updates on the job's own "
                    "first batches (default 8: within one source forward of the timed size)")
    ap.add_argument("--gc", default="lap", choices=("lap", "auto"),
                    help="lap (default): Python's cyclic garbage collector is switched off while a job runs and called once "
                         "per job where the host has nothing to do -- while the batched LAP kernel runs; auto: the "
                         "interpreter's own schedule (a full collection over the job's fx graphs / tap dictionaries then "
                         "lands inside every other job: 6.36 / 6.52 s alternating)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank logic on a single GPU)")
    ap.add_argument("--all-ranks-on-gpu0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="profiling aid, single process: time rank 0's share of an N-rank job with the collectives "
                         "skipped (PLEAS_EMULATE_WORLD); the line is tagged emulated and is NOT a benchmark result")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start ``torch.distributed.run`` with N ranks as a CHILD process
    (this process has not touched the GPU and never will) and return its exit status.  Rank 0 prints the JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // max(1, args.gpus))))
    return subprocess.call(cmd, env=env)


def source_conv_mode():
    from pleas_merging_amd.methods import source_forward

    return source_forward.SOURCE_CONV


def usable_cores():
    """Host cores this process may really use (affinity mask and cgroup quota, not the socket count)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _early = parse()
    if _early.gpus > 1:          # before `import torch`: the parent stays free of any GPU state
        sys.exit(self_launch(_early))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


class Pool:
    """Synthetic batches: batch b = N(0,1) from a generator seeded 1000 + b, generated on the device.  ``where`` =
    "resident": the loops are handed the HBM-resident tensors; "host": PINNED host copies -- every batch then crosses PCIe
    inside the timed region, as in the reference's loops (activation_matching.py:121, pleas_merging.py:266), on a copy
    stream beside the previous batch's compute (hip_ops.to_device_async)."""

    def __init__(self, n, batch, device, host=False):
        gen = torch.Generator(device=device)
        self.items, self.host_items = [], []
        for b in range(n):
            gen.manual_seed(1000 + b)
            x = torch.randn(batch, 3, 224, 224, generator=gen, device=device)
            self.items.append(x)
            if host:
                h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
                h.copy_(x)
                self.host_items.append(h)
        self.where = "resident"

    def loader(self, start, count):
        src = self.host_items if self.where == "host" else self.items
        return [(src[(start + i) % len(src)], None) for i in range(count)]


def build_models(arch, device, batch):
    from pleas_merging_amd import resnet as zoo

    models = []
    gen = torch.Generator(device=device)
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = zoo.MODELS[arch](num_classes=1000).to(device)
        calib = []
        for i in range(4):
            gen.manual_seed(900 + i)
            calib.append(torch.randn(batch, 3, 224, 224, generator=gen, device=device))
        zoo.calibrate_bn(m, calib)
        models.append(m)
    return models


class Phases:
    """Wall-clock of the job's phases.  Only the extra job after the timed ones uses it: every boundary synchronises the
    device, which removes the overlap between phases (LAP beside prefetched source forwards), so the phases add up to
    slightly more than a timed job."""

    def __init__(self, on=False, verbose=False):
        self.on, self.verbose, self.t, self.out = on, verbose, time.perf_counter(), {}

    def mark(self, name):
        if not self.on:
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.out[name] = round(self.out.get(name, 0.0) + now - self.t, 4)
        if self.verbose:
            log("phase %-28s %.3f s" % (name, now - self.t))
        self.t = now


def run_job(cfg, spec, m1, m2, match_loader, pleas_loader, n_pleas_sched, phases=None, after_matching=None):
    """The timed hot path: activation matching (+ LAP) -> partial merge -> PLeaS updates.  Returns a dict with the merged
    model, the permutation, the costs and the per-layer losses of the first and last update (device tensors).
    ``after_matching(perm, costs) -> (perm, costs)`` (parity leg only, never in a timed job): lets the checker look at the
    job's assignment and hand back the one the merge and the updates shall continue from."""
    from pleas_merging_amd.core.solvers import hip_solve_lsa
    from pleas_merging_amd import hip_ops
    from pleas_merging_amd.methods.activation_matching import accumulate_costs_fused, activation_matching, solve_all
    from pleas_merging_amd.methods.partial_matching import partial_merge
    from pleas_merging_amd.methods.pleas_merging import FrozenSources, PleasFitter

    phases = phases or Phases()
    dp = cfg["dp"]
    if cfg.get("gc") == "lap":
        gc.disable()
    if cfg.get("match_mode") == "train":      # the drivers' mode (the PLeaS phase puts the sources back into eval mode)
        m1.train()
        m2.train()
    # The frozen sources of the PLeaS phase do not depend on the permutation: while the batched LAP kernel runs (one
    # workgroup per problem), the host builds their fused forwards and enqueues the first groups of source forwards on
    # the side streams, where they also fill the GPU's idle time during partial merge and fitter set-up.
    early = {}
    inputs = [x for x, _ in pleas_loader]
    closed_form = cfg.get("solver") == "normal_eq"
    rank, world = (0, 1)
    if closed_form and dp:      # whole batches per rank (b % world == rank), never sample slices
        from pleas_merging_amd.methods.activation_matching import _dist_info

        rank, world = _dist_info()
        inputs = inputs[rank::world]
    per_forward = cfg.get("closed_form_per_forward") if closed_form else cfg["sources_per_forward"]

    def while_solving():
        early["sources"] = src = FrozenSources(m1, m2, data_parallel=dp and not closed_form)
        src.prefetch(inputs, group=per_forward, max_groups=cfg["prefetch_groups"],
                     memory_fraction=cfg["prefetch_memory"])
        if cfg.get("gc") == "lap":      # the LAP kernel has ~0.2 s to go and the host nothing to enqueue: collect now
            gc.collect()

    if phases.on:      # the same two calls activation_matching() makes, with a synchronising boundary between them
        costs = accumulate_costs_fused(spec, m1, m2, match_loader, len(match_loader), hip_ops.EPI_NEG_CDIST,
                                       batches_per_forward=cfg["match_per_forward"])
        phases.mark("matching")
        perm = solve_all(costs, hip_solve_lsa, while_solving)
        phases.mark("lap")
    else:
        perm, costs = activation_matching(spec, m1, m2, match_loader, len(match_loader), output_costs=True,
                                          while_solving=while_solving, batches_per_forward=cfg["match_per_forward"])
    hip_perm, hip_costs = perm, costs
    if after_matching is not None:
        perm, costs = after_matching(perm, costs)
    m3 = partial_merge(spec, m1, m2, perm, costs, cfg["ratio"], device=next(m1.parameters()).device)   # stays on the GPU
    # Data parallel: each rank's share of an update is small (batch / world samples); the frozen sources therefore forward
    # 2 * world updates' samples at once (steps() default), which keeps their host dispatch off the per-update path.
    solve_info = None
    if closed_form:
        from pleas_merging_amd.methods.normal_eq import NormalEqFitter

        fit = NormalEqFitter(m1, m2, m3, spec, perm, costs, cfg["ratio"], n_pleas_sched, sources=early.get("sources"))
        fit.rank, fit.world = rank, world
        phases.mark("merge_and_setup")
        for _ in fit.steps(inputs, sources_per_forward=per_forward):
            pass
        phases.mark("updates")
        solve_info = fit.solve()
        first = last = None
    else:
        fit = PleasFitter(m1, m2, m3, spec, perm, costs, cfg["ratio"], n_pleas_sched, data_parallel=dp,
                          sources=early.get("sources"), shard_optimizer=cfg["shard_optimizer"])
        phases.mark("merge_and_setup")
        first = None
        for i in fit.steps(inputs, lookahead=cfg["lookahead"], sources_per_forward=cfg["sources_per_forward"]):
            if i == 0:
                first = fit.loss_now.clone()
        last = fit.loss_now.clone()
        phases.mark("updates")
    m3 = fit.finish()
    phases.mark("finish")
    if cfg.get("gc") == "lap":
        gc.enable()
    return {"m3": m3, "perm": perm, "costs": costs, "first_loss": first, "last_loss": last, "layers": len(fit.plans),
            "hip_perm": hip_perm, "hip_costs": hip_costs, "solve_info": solve_info}


def check_result(spec, res, full=True):
    """Cheap invariants of a job's result (no oracle at this size inside the bench; tests/test_hip_fullsize.py compares
    the same calls with the CPU oracle at batch 2)."""
    out = {}
    perms_ok = all(sorted(res["perm"][k].tolist()) == list(range(spec[k].size)) for k in spec)
    out["perms_are_permutations"] = bool(perms_ok and len(res["perm"]) == len(spec))
    out["costs_finite"] = bool(all(torch.isfinite(v).all().item() for v in res["costs"].values()))
    out["weights_finite"] = bool(all(torch.isfinite(v).all().item() for v in res["m3"].state_dict().values()
                                     if v.dtype.is_floating_point))
    if res["first_loss"] is None:      # closed form: no per-update losses
        out["fp64_fallbacks"] = int((res.get("solve_info") or {}).get("fp64_fallbacks", 0))
        out["ok"] = bool(perms_ok and out["costs_finite"] and out["weights_finite"])
        return out
    first, last = res["first_loss"].double().cpu(), res["last_loss"].double().cpu()
    out["loss_first_update"] = float(first.sum())
    out["loss_last_update"] = float(last.sum())
    # the stem's loss is rounding noise; every other layer must have come down
    fell = int((last[1:] < first[1:]).sum())
    out["layers_whose_loss_fell"] = "%d / %d" % (fell, first.numel() - 1)
    out["weights_finite"] = bool(all(torch.isfinite(v).all().item() for v in res["m3"].state_dict().values()
                                     if v.dtype.is_floating_point))
    out["ok"] = bool(out["perms_are_permutations"] and out["costs_finite"] and out["weights_finite"])
    if full:   # a handful of updates of a shortened debug job need not bring every layer's loss down
        out["ok"] = bool(out["ok"] and out["loss_last_update"] < out["loss_first_update"] and fell >= 0.9 * (first.numel() - 1))
    return out


def time_normal_eq(spec, m1, m2, perm, costs, loader, ratio, sources_per_forward=None):
    """Closed-form alternative to the Adam phase on the same batches: accumulate A, B^T, then solve."""
    from pleas_merging_amd import hip_ops
    from pleas_merging_amd.methods.normal_eq import NormalEqFitter
    from pleas_merging_amd.methods.partial_matching import partial_merge

    m3 = partial_merge(spec, m1, m2, perm, costs, ratio)
    fit = NormalEqFitter(m1, m2, m3, spec, perm, costs, ratio, len(loader))
    fit.step(loader[0][0])  # warm-up (plans, workspaces); its contribution is removed again
    fit.A_flat.zero_()
    fit.B_flat.zero_()
    for st in fit.bias_stats.values():
        for t in st:
            t.zero_()
    hip_ops.profile_reset()
    hip_ops.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in fit.steps((x for x, _ in loader), sources_per_forward=sources_per_forward):   # same source-forward size as the jobs
        pass
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    solve_info = fit.solve()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hip_ops.profile_enable(False)
    p = hip_ops.profile_collect()
    plan = fit.neq.plan_info()
    fit.finish()
    rec = p.get("normal_eq", (0, 0.0, 0.0, 0.0))
    ach = rec[2] / (rec[1] * 1e-3) / 1e12 if rec[1] > 0 else 0.0
    executed = plan["flops_executed"] * rec[0] / (rec[1] * 1e-3) / 1e12 if rec[1] > 0 else 0.0
    # systems for the CPU leg (fp64 lstsq on the SAME accumulated A / B): the smallest, the most common and the largest
    # 3x3 layer; fit.A keeps the accumulated lower triangle (the solve factorises copies)
    samples, seen = [], set()
    for idx, pl in enumerate(fit.plans):
        k = fit.K[idx]
        if ratio == 0.0 and k in (576, 2304, 4608) and k not in seen and len(pl.w_shape) == 4 and pl.w_shape[2] == 3 \
                and pl.b is None:
            seen.add(k)
            A = fit.A[idx]
            w = pl.w.detach().reshape(pl.w_shape[0], k)        # kernel-position-major rows, as B^T is
            samples.append({"name": pl.name, "K": k, "cout": pl.w_shape[0], "ridge": fit.ridge,
                            "A": (torch.tril(A) + torch.tril(A, -1).t()).cpu(), "Bt": fit.Bt[idx].cpu(), "W_hip": w.cpu()})
    return {"solver": "normal_eq", "batches": len(loader), "accumulate_s": round(t1 - t0, 3), "solve_s": round(t2 - t1, 3),
            "fp64_fallbacks": int(solve_info.get("fp64_fallbacks", 0)),
            "fp64_fallbacks_note": "systems the fp32 HIP Cholesky flagged (non-positive pivot) and torch.linalg redid in fp64",
            "kernels_ms": {k: {"launches": v[0], "total_ms": round(v[1], 2)} for k, v in sorted(p.items()) if v[0]},
            "kernels_note": "own kernels of this leg by HIP events (on several streams: they overlap); what is left of accumulate_s is "
                            "what the events do not bracket (the classifier's GEMM, elementwise torch ops), the bias statistics and host dispatch",
            "neq_batch_kernel": {"bound": "mfma", "launches": rec[0], "avg_launch_us": round(rec[1] * 1e3 / max(rec[0], 1), 1),
                                 "achieved": round(ach, 1), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(ach / FP32_MATRIX_PEAK_TFLOPS, 3),
                                 "algorithmic_flop_per_launch": plan["flops"], "executed_flop_per_launch": plan["flops_executed"],
                                 "executed_tflops": round(executed, 1), "executed_frac": round(executed / FP32_MATRIX_PEAK_TFLOPS, 3),
                                 "note": "`achieved` = the path's flops (lower block triangle, K^2 * N*HWo per layer) / time; "
                                         "`executed_*` = what the grid really multiplies: stride-1 3x3 layers contract one "
                                         "block per lag class (29 of 45, the rest are copies made once at the end), whole "
                                         "tiles counted -- that one is the MFMA utilisation"},
            "_samples": samples, "_K": list(fit.K)}


BF16_MATRIX_PEAK_TFLOPS = 2516.6      # MI355X_MICROARCH.md: 16 x the fp32 matrix rate (~2.5 PFLOP/s dense)


def time_alt_arith(job, res, spec, value, n_match, world, rank, args):
    """The SAME job under ``pleas_arith(PLEAS_ARITH_SPLIT_BF16)``: every fp32 operand of the contraction
    kernels as the exact sum of three bf16 values, six bf16-MFMA products per k step, fp32 accumulation -- fp32 accuracy at 2.67x
    less matrix-pipe time.  One untimed job, two timed ones; the kernels' HIP-event times; the result against the fp32-MFMA
    job's on the same batches (assignments, costs, trained weights).  NOT the headline: ``value`` / ``dtype`` stay on the exact
    fp32 MFMA path; ``parity_vs_oracle`` / ``parity_vs_fp64`` of this arithmetic are added by the cpu_baseline leg."""
    from pleas_merging_amd import _lib, hip_ops

    lib = _lib.lib()
    lib.pleas_arith(1)
    try:
        job()
        torch.cuda.synchronize()
        hip_ops.profile_reset()
        hip_ops.profile_enable(True, skip=("bn_act",))
        times, alt = [], None
        for _ in range(2):
            t0 = time.perf_counter()
            alt = job()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        hip_ops.profile_enable(False)
        prof = hip_ops.profile_collect()
    finally:
        lib.pleas_arith(0)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    same = sum(1 for k in spec if torch.equal(alt["perm"][k], res["perm"][k]))
    # a near-tie group assigned differently merges its tensors differently: those are two different (equally good) jobs from there
    # on, so the weight comparison leaves out every tensor that carries an axis of such a group (layers are fitted independently)
    flipped = {ax.key for k in spec if not torch.equal(alt["perm"][k], res["perm"][k]) for ax in spec[k].state}
    flipped |= {key.rsplit(".", 1)[0] + ".bias" for key in flipped}
    sd_a, sd_b = alt["m3"].state_dict(), res["m3"].state_dict()
    rels = {k: rel(v, sd_b[k]) for k, v in sd_a.items() if v.dtype.is_floating_point and k != "conv1.weight" and k not in flipped}
    worst = max(rels, key=rels.get)
    kernels = {}
    for name, label in (("gram_partial", "gram_batch_kernel"), ("conv_fwd", "fwd_batch_kernel"), ("conv_wgrad", "wgrad_batch_kernel"),
                        ("conv2d", "conv2d_fwd_kernel (k x k source convolutions)")):
        rec = prof.get(name, (0, 0.0, 0.0, 0.0))
        if rec[0]:
            tf = rec[2] / (rec[1] * 1e-3) / 1e12
            kernels[name] = {"kernel": label, "launches": rec[0], "avg_launch_us": round(rec[1] * 1e3 / rec[0], 2),
                             "fp32_equivalent_tflops": round(tf, 1), "of_fp32_matrix_peak": round(tf / FP32_MATRIX_PEAK_TFLOPS, 3),
                             "of_bf16_matrix_peak_over_6": round(tf / (BF16_MATRIX_PEAK_TFLOPS / 6.0), 3)}
    out = {"arithmetic": "split bf16: x = h1 + h2 + h3 exactly, 6 of the 9 bf16 products per fp32 product on v_mfma_f32_32x32x16_bf16, "
                         "fp32 accumulation (pleas_arith(PLEAS_ARITH_SPLIT_BF16)); tile forms without a split variant stay exact",
           "job_s": round(min(times), 4), "jobs": [round(t, 4) for t in times], "speedup_vs_value": round(value / min(times), 3),
           "kernels": kernels,
           "vs_fp32_mfma_job": {"assignments_identical": "%d / %d" % (same, len(spec)),
                                "worst_group_cost_rel_fro": max(rel(alt["costs"][k], res["costs"][k]) for k in spec),
                                "worst_trained_tensor": worst, "worst_trained_rel_fro": rels[worst],
                                "tensors_above_1e-4": sum(1 for v in rels.values() if v > 1e-4), "tensors_compared": len(rels),
                                "tensors_on_a_differently_assigned_near_tie_group": len([k for k in sd_a if k in flipped]),
                                "note": "full jobs on the same batches; weights after 401 Adam updates (sign-like first steps "
                                        "amplify any rounding difference, as between any two fp32 implementations); tensors "
                                        "that carry an axis of a differently assigned near-tie group are left out"},
           "note": "NOT the headline: value / dtype stay on the exact fp32 MFMA path until this leg has been through a green "
                   "driver GPUTEST (tests/test_hip_split_bf16.py)"}
    log("alt_arith (split bf16): %.3f s per job (fp32 MFMA %.3f), assignments identical %d / %d" % (min(times), value, same, len(spec)))
    return out


def time_sources_alone(m1, m2, pool, n_updates, per=2, groups=4):
    """The frozen source forwards of the PLeaS phase (vendor convolutions + pleas_bn_act, both models on two streams,
    ``per`` updates' batches per forward as in the jobs) with nothing else on the GPU, scaled to the job's updates."""
    from pleas_merging_amd.methods.pleas_merging import FrozenSources

    src = FrozenSources(m1, m2)
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in range(groups):
            src.launch_group([pool.items[(per * g + i) % len(pool.items)] for i in range(per)])
            src.queue.clear()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    src.close()
    per_update = dt / (groups * per)
    return per_update * n_updates, per_update


def time_bn_reset(m3, pool, n_batches=101):
    """The step right after ``train`` in every reference driver (run_domainnet.py:327-341): 101 train-mode forwards of the
    merged model that recompute its BatchNorm statistics.  HIP BatchNorm path (``pleas_bn_train_fold`` + ``pleas_bn_act``
    through the fx rewrite) and the vendor modules, same batches, one untimed pass each first."""
    import copy

    from pleas_merging_amd.methods.extras import reset_bn_stats

    loader = pool.loader(0, n_batches)
    out = {}
    for name, fused in (("hip_s", True), ("vendor_modules_s", False)):
        model = copy.deepcopy(m3)
        reset_bn_stats(model, loader[:8], 8, fused=fused)      # two forwards of the timed size on the HIP path
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reset_bn_stats(model, loader, n_batches, fused=fused)
        torch.cuda.synchronize()
        out[name] = round(time.perf_counter() - t0, 3)
        del model
    out["note"] = ("BN-statistics reset of the merged model, %d batches of %d (not in `value`: the reference's `train` ends "
                   "before it); hip = four batches per forward, statistics folded per batch on the device (one launch per "
                   "BatchNorm and forward) + one bn_act pass per chain" % (n_batches, pool.items[0].shape[0]))
    return out


def cpu_baseline_and_parity(cfg, spec, m1, m2, pool, n_match, n_pleas, n_sched, sample_match, sample_updates, alt_arith=False):
    """The CPU oracle (restatement of the reference path) on this box's host cores, on the job's OWN first batches at the
    job's batch size -- ``sample_match`` matching batches and ``sample_updates`` PLeaS updates, i.e. the count is
    subsampled, not the batch -- and, on exactly those batches, the HIP job with the timed knobs (same batches per twin
    forward, same updates per source forward, prefetch during the LAP) compared with it:
      * worst group-cost rel-fro, assignments equal k / groups (a differing group must be a near tie: the HIP assignment
        is what the oracle's LAP returns on the HIP costs and its value under the ORACLE's costs is within 1e-6);
      * merged state dict bit-equal; trained tensors after the sample's updates (both sides continue from the ORACLE's
        assignment) against the oracle, with the oracle's own oneDNN-on / off disagreement as the yardstick
        (tests/test_hip_timed_config.py is the same comparison as a test);
      * the same sample in fp64 (oracle.fp64_anchor): distances TO it are statements about accuracy.
    ``alt_arith``: the HIP job once more under pleas_arith(PLEAS_ARITH_SPLIT_BF16), compared with the SAME oracle results.
    Returns (cpu_baseline, parity, parity of the split-bf16 job or None)."""
    import copy

    from oracle import pleas_oracle as orc

    cores = usable_cores()
    torch.set_num_threads(cores)
    nM, nU = min(n_match, sample_match), min(n_pleas, sample_updates)
    log("cpu baseline + parity on %d cores: %d matching batches + %d updates of the job's own batches" % (cores, nM, nU))
    c1, c2 = copy.deepcopy(m1).cpu(), copy.deepcopy(m2).cpu()
    data = [(x.cpu(), None) for x, _ in pool.loader(0, max(nM, nU))]
    ratio = cfg["ratio"]
    seen = {}

    def after_matching(perm, costs):
        if "want_perm" not in seen:
            if cfg.get("match_mode") == "train":      # the oracle matches in the drivers' mode too (its `train` goes back to eval)
                c1.train()
                c2.train()
            t0 = time.time()
            want_costs = orc.matching_costs(spec, c1, c2, data[:nM], nM, accumulate=True)
            seen["t_match"] = (time.time() - t0) / nM
            t0 = time.time()
            want_perm = {k: orc.solve_lsa(v) for k, v in want_costs.items()}
            seen["t_lap"] = time.time() - t0
            log("cpu: matching batch %.1fs, all LAPs %.2fs" % (seen["t_match"], seen["t_lap"]))
            seen["want_perm"], seen["want_costs"] = want_perm, want_costs
        return seen["want_perm"], {k: v.to(costs[k].device) for k, v in seen["want_costs"].items()}

    res = run_job(cfg, spec, m1, m2, pool.loader(0, nM), pool.loader(0, nU), n_sched, after_matching=after_matching)
    torch.cuda.synchronize()
    res_alt = None
    if alt_arith:
        from pleas_merging_amd import _lib

        _lib.lib().pleas_arith(1)
        try:
            res_alt = run_job(cfg, spec, m1, m2, pool.loader(0, nM), pool.loader(0, nU), n_sched, after_matching=after_matching)
            torch.cuda.synchronize()
        finally:
            _lib.lib().pleas_arith(0)
    want_perm, want_costs = seen["want_perm"], seen["want_costs"]
    rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-30))
    value = lambda cost, perm: float(cost.double().cpu()[torch.arange(len(perm)), perm].sum())
    # ---- merge + updates on the CPU, from the same assignment
    t0 = time.time()
    o3 = orc.partial_merge(spec, c1, c2, want_perm, want_costs, ratio)
    t_merge = time.time() - t0
    merged = {k: v.clone() for k, v in o3.state_dict().items()}
    t0 = time.time()
    o3, _ = orc.train(data[:nU], c1, c2, o3, spec, want_perm, want_costs, ratio, n_sched)
    t_step = (time.time() - t0) / nU
    log("cpu: PLeaS update %.1fs" % t_step)
    want = o3.state_dict()
    with torch.backends.mkldnn.flags(enabled=False):      # the oracle against itself: the yardstick at this depth
        v3 = orc.partial_merge(spec, c1, c2, want_perm, want_costs, ratio)
        v3, _ = orc.train(data[:nU], c1, c2, v3, spec, want_perm, want_costs, ratio, n_sched)
    variant = v3.state_dict()
    # ---- the fp64 ANCHOR on the same batches: how far is each fp32 implementation from the exact path?  (distances between two
    # fp32 runs -- the yardstick above -- are statements about spread; these are statements about accuracy)
    # (drivers' train-mode matching moves the running statistics batch by batch: the cost anchor is taken in eval mode only)
    t0 = time.time()
    costs64, want64 = orc.fp64_anchor(spec, c1, c2, data, nM if cfg.get("match_mode") != "train" else 0, nU, want_perm,
                                      want_costs, ratio, n_sched)
    log("cpu: fp64 anchor %.1fs" % (time.time() - t0))

    def compare(res):
        worst_cost = max(rel(res["hip_costs"][k], want_costs[k]) for k in spec)
        equal, near_ties, bad_groups = 0, {}, []
        for k in spec:
            if (res["hip_perm"][k] == want_perm[k]).all():
                equal += 1
                continue
            same_lap = bool((orc.solve_lsa(res["hip_costs"][k].cpu()) == res["hip_perm"][k]).all())
            best, mine = value(want_costs[k], want_perm[k]), value(want_costs[k], res["hip_perm"][k])
            gap = (best - mine) / abs(best)
            near_ties[str(k)] = [int((res["hip_perm"][k] != want_perm[k]).sum()), gap]
            if not (same_lap and 0 <= gap < 1e-6):
                bad_groups.append(str(k))
        got = {k: v.cpu() for k, v in res["m3"].state_dict().items()}
        rows = {}
        for k in want:
            if k == "conv1.weight" or not want[k].dtype.is_floating_point or torch.equal(want[k], merged[k]):
                continue      # the stem's residual is rounding noise in the reference itself
            rows[k] = (rel(got[k], want[k]), rel(variant[k], want[k]))
        worst = max(rows, key=lambda k: rows[k][0]) if rows else None
        yard_max = max((v[1] for v in rows.values()), default=0.0)
        over = {k: v for k, v in rows.items() if v[0] > max(1e-4, 3 * v[1])}
        # tensors above 3 x their own yardstick: 0-1 of 105 observed in eval-mode matching, 2 in the drivers' train mode (one of them
        # at 1.006e-4 against a 1e-4 floor); a wrong kernel moves dozens
        ok = (worst_cost < 1e-4 and not bad_groups and len(near_ties) <= 4 and len(over) <= max(2, len(rows) // 50)
              and all(v[0] <= max(1e-4, 3 * yard_max) for v in over.values()))
        parity = {
            "ok": bool(ok), "matching_batches": nM, "updates": nU, "batch": int(data[0][0].shape[0]),
            "worst_group_cost_rel_fro": worst_cost, "assignments_equal": "%d / %d" % (equal, len(spec)),
            "near_tie_groups": near_ties, "groups_that_are_not_near_ties": bad_groups,
            "trained_tensors": len(rows), "worst_trained_tensor": worst,
            "worst_trained_rel_fro": rows[worst][0] if worst else None,
            "oracle_self_spread_of_that_tensor": rows[worst][1] if worst else None, "oracle_self_spread_worst": yard_max,
            "tensors_above_1e-4": sum(1 for v in rows.values() if v[0] > 1e-4),
            "tensors_above_3x_own_yardstick": {k: list(v) for k, v in over.items()},
            "note": "HIP job with the timed knobs on the job's own first batches vs the CPU oracle on the same batches; trained "
                    "tensors: both sides continue from the oracle's assignment, yardstick = the oracle with oneDNN "
                    "convolutions on vs off; gate as tests/test_hip_timed_config.py"}
        cost_rows = {str(k): (rel(res["hip_costs"][k], costs64[k]), rel(want_costs[k], costs64[k])) for k in spec} \
            if costs64 is not None else {"-": (0.0, 0.0)}
        t_rows = {k: (rel(got[k], want64[k]), max(rel(want[k], want64[k]), rel(variant[k], want64[k]))) for k in rows}
        ratios = sorted(a / max(b, 1e-30) for a, b in t_rows.values()) or [0.0]
        ref_max = max((b for _, b in t_rows.values()), default=0.0)
        over_c = {k: list(v) for k, v in cost_rows.items() if v[0] > max(2e-6, 1.5 * v[1])}
        over_t = {k: list(v) for k, v in t_rows.items() if v[0] > max(1e-4, 1.5 * v[1])}
        anchor_ok = (not over_c and ratios[len(ratios) // 2] <= 1.25 and len(over_t) <= max(1, len(t_rows) // 20)
                     and all(v[0] <= max(1e-4, 3 * ref_max) for v in over_t.values()))
        parity["ok"] = bool(parity["ok"] and anchor_ok)
        parity["parity_vs_fp64"] = {
            "ok": bool(anchor_ok), "groups": len(cost_rows), "worst_group_cost_hip_vs_fp64": max(a for a, _ in cost_rows.values()),
            "worst_group_cost_oracle_fp32_vs_fp64": max(b for _, b in cost_rows.values()), "cost_groups_above_1.5x": over_c,
            "tensors": len(t_rows), "worst_tensor_hip_vs_fp64": max((a for a, _ in t_rows.values()), default=0.0),
            "worst_tensor_oracle_fp32_vs_fp64": ref_max, "median_ratio_hip_over_oracle_fp32": ratios[len(ratios) // 2],
            "tensors_above_1e-4_vs_fp64": {"hip": sum(1 for a, _ in t_rows.values() if a > 1e-4),
                                            "oracle_fp32": sum(1 for _, b in t_rows.values() if b > 1e-4)},
            "tensors_above_1.5x_own_oracle_distance": over_t,
            "note": "same batches in fp64 (oracle.fp64_anchor); per group cost hip <= 1.5 x oracle-fp32 distance (floor 2e-6); trained "
                    "tensors: median ratio <= 1.25, at most 1 in 20 above 1.5 x its own oracle distance (floor 1e-4; larger of the "
                    "oneDNN on / off variants), none above 3 x the model's largest -- the gate of tests/test_hip_timed_config.py"}
        return parity

    parity = compare(res)
    parity_alt = compare(res_alt) if res_alt is not None else None
    total = n_match * seen["t_match"] + seen["t_lap"] + t_merge + n_pleas * t_step
    cpu = {"value": round(total, 1), "unit": "s", "cores": cores, "kind": "port",
           "sample": "oracle on the job's own batches at batch %d: %d matching batches (%.1fs each) + %d PLeaS updates (%.1fs "
                     "each), all %d LAPs (%.2fs), merge (%.2fs); counts scaled to %d matching batches + %d updates"
                     % (data[0][0].shape[0], nM, seen["t_match"], nU, t_step, len(want_perm), seen["t_lap"], t_merge, n_match,
                        n_pleas)}
    return cpu, parity, parity_alt


class _NodeTap(torch.fx.Interpreter):
    """Keeps a tracked node's value at the moment it is produced (before a later in-place ReLU can overwrite it), which is
    when the reference's cross module evaluates it (activation_matching.py:90-92)."""

    def __init__(self, gm, wanted):
        super().__init__(gm)
        self.wanted, self.kept = set(wanted), {}

    def run_node(self, n):
        out = super().run_node(n)
        if n.name in self.wanted and torch.is_tensor(out):
            self.kept[n.name] = out.clone()
        return out


def library_baseline(cfg, spec, m1, m2, pool, n_match, n_pleas, n_sched, sample_match, sample_updates, perm, costs):
    """The reference's loops on THIS GPU with stock PyTorch-ROCm operators -- what the reference itself would do on the card
    (SURVEY.md 8(d) last row), beside the own-kernel job: per tracked node ``-torch.cdist`` of the two
    models' activations, summed per node and per group (activation_matching.py:31-46, :119-134); D2H + scipy's LAP
    (core/solvers.py:29-31); per update two hooked source forwards, per layer ``l1(ip1)``, ``l2(ip2)``, the index_select /
    cat assembly of (ip, op), ``((layer(ip) - op) ** 2).mean()``, one backward, the gradient mask, ``torch.optim.Adam`` and
    the cosine schedule (pleas_merging.py:63-149, :262-291, :358-375).  Vendor kernels only (MIOpen, rocBLAS / Tensile, ATen);
    only the one-off partial merge (merge_s) is the library's.  Timed on a bounded sample of the job's own batches after one
    untimed pass (MIOpen's first-use work), counts scaled to the job -- as ``cpu_baseline`` is."""
    import copy

    import torch.nn as nn
    from scipy.optimize import linear_sum_assignment

    from pleas_merging_amd.core.utils import Axis
    from pleas_merging_amd.methods.partial_matching import get_blocks, partial_merge, spread_blocks

    dev = next(m1.parameters()).device
    nM, nU = min(n_match, sample_match), min(n_pleas, sample_updates)
    tracked = [ax for pg in spec.values() for ax in pg.node]
    names = {ax.key for ax in tracked}
    m1.eval()
    m2.eval()
    taps = [_NodeTap(torch.fx.symbolic_trace(m), names) for m in (m1, m2)]

    def cross(x, y, a):
        xf = torch.movedim(x, a, 0).reshape(x.shape[a], -1)
        yf = torch.movedim(y, a, 0).reshape(y.shape[a], -1)
        return -torch.cdist(xf[None], yf[None])[0]

    def matching(loader):
        per_node = {}
        with torch.inference_mode():
            for x, _ in loader:
                x = x.to(dev)
                for t in taps:
                    t.kept = {}
                    t.run(x)
                for ax in tracked:
                    val = cross(taps[0].kept[ax.key], taps[1].kept[ax.key], ax.axis)
                    per_node[ax] = per_node[ax] + val if ax in per_node else val
        out = {}
        for key, pg in spec.items():
            total = 0
            for nax in pg.node:
                total = total + per_node[nax]
            out[key] = total
        return out

    matching(pool.loader(0, 1))      # untimed: first use of every operator configuration
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib_costs = matching(pool.loader(0, nM))
    torch.cuda.synchronize()
    t_match = (time.perf_counter() - t0) / nM
    t0 = time.perf_counter()
    lib_perm = {}
    for k, v in lib_costs.items():      # the reference's solver: D2H, scipy, maximize
        rows, cols = linear_sum_assignment(v.detach().cpu().numpy(), maximize=True)
        lib_perm[k] = torch.from_numpy(cols)
    t_lap = time.perf_counter() - t0
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    worst_cost = None
    if nM == n_match:      # the own-kernel job's costs cover the same batches only when nothing was subsampled
        worst_cost = max(rel(lib_costs[k], costs[k]) for k in spec)
    del lib_costs, taps
    # ---- PLeaS updates, the reference's step() with stock autograd
    t0 = time.perf_counter()
    m3 = partial_merge(spec, m1, m2, perm, costs, cfg["ratio"], device=dev)
    t_merge = time.perf_counter() - t0
    blocks = spread_blocks(spec, get_blocks(spec, perm, costs, cfg["ratio"], False))
    layers = {n: copy.deepcopy(m).to(dev) for n, m in m3.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    src1, src2 = dict(m1.named_modules()), dict(m2.named_modules())
    params, masks = [], []
    for name, layer in layers.items():
        bi, bo = blocks.get(Axis(name + ".weight", 1)), blocks.get(Axis(name + ".weight", 0))
        ni, mi = (len(bi[0]), len(bi[2])) if bi is not None else (3, 0)
        no, mo = (len(bo[0]), len(bo[2])) if bo is not None else (1000, 0)
        for prm in layer.parameters():
            prm.requires_grad_(True)
            params.append(prm)
            mk = torch.ones_like(prm)
            if prm.dim() >= 2:      # the reference's transposed indexing (pleas_merging.py:57-58)
                mk[ni:ni + mi, no + mo:no + 2 * mo] = 0.0
                mk[ni + mi:ni + 2 * mi, no:no + mo] = 0.0
            masks.append(mk)
    opt = torch.optim.Adam(params, lr=5e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, n_sched)
    acts1, acts2, handles = {}, {}, []
    for model, store in ((m1, acts1), (m2, acts2)):
        for name, mod in model.named_modules():
            if isinstance(mod, (nn.Conv2d, nn.Linear, nn.LayerNorm)):
                handles.append(mod.register_forward_hook(lambda m, i, o, name=name, store=store: store.__setitem__(name, i[0])))
    sel = lambda t, idx: t.index_select(1, idx)

    def update(x):
        with torch.no_grad():
            m1(x)
            m2(x)
        opt.zero_grad()
        total = 0.0
        for name, layer in layers.items():
            with torch.no_grad():
                ip1, ip2 = acts1[name], acts2[name]
                bo, bi = blocks.get(Axis(name + ".weight", 0)), blocks.get(Axis(name + ".weight", 1))
                if bo is None:
                    w = torch.arange(1000, device=dev)
                    bo = (w, w, w[:0], w[:0])
                if bi is None:
                    w = torch.arange(ip1.shape[1], device=dev)
                    bi = (w, w, w[:0], w[:0])
                o1, o2 = src1[name](ip1), src2[name](ip2)
                ip = torch.cat([(sel(ip1, bi[0]) + sel(ip2, bi[1])) / 2, sel(ip1, bi[2]), sel(ip2, bi[3])], 1)
                op = torch.cat([(sel(o1, bo[0]) + sel(o2, bo[1])) / 2, sel(o1, bo[2]), sel(o2, bo[3])], 1)
            total = total + ((layer(ip) - op) ** 2).mean()
        total.backward()
        for prm, mk in zip(params, masks):
            prm.grad *= mk
        opt.step()
        sched.step()
        acts1.clear()
        acts2.clear()
        return total.detach()

    xs = [x.to(dev) for x, _ in pool.loader(0, nU)]
    update(xs[0])      # untimed (first use of every backward configuration); its step is part of the state, as in any warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in xs:
        last = update(x)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / nU
    for h in handles:
        h.remove()
    total = n_match * t_match + t_lap + t_merge + n_pleas * t_step
    log("library baseline (stock PyTorch-ROCm ops on this GPU): matching batch %.3fs, scipy LAPs %.2fs, update %.3fs -> %.1fs per job"
        % (t_match, t_lap, t_step, total))
    return {"value": round(total, 2), "unit": "s", "kind": "stock PyTorch-ROCm operators on the same GPU (torch.cdist, MIOpen / "
            "Tensile convolutions under autograd, torch.optim.Adam, scipy LAP after D2H): the reference's own loops",
            "matching_batch_s": round(t_match, 4), "scipy_lap_s": round(t_lap, 3), "merge_s": round(t_merge, 3),
            "update_s": round(t_step, 4), "loss_last_update": float(last),
            "worst_group_cost_rel_fro_vs_own_kernels": worst_cost,
            "sample": "%d matching batches + %d updates of the job's own batches at batch %d after one untimed pass each, all %d "
                      "LAPs; counts scaled to %d + %d" % (nM, nU, xs[0].shape[0], len(lib_perm), n_match, n_pleas)}


def cpu_reference_legs(costs, hip_perm, alt):
    """north_star's CPU path as worded -- "scipy LAP + torch.linalg.lstsq" -- on this box's host cores:
    * ``scipy.optimize.linear_sum_assignment(maximize=True)`` (the reference's solver, core/solvers.py:29-31) on the 71 cost
      matrices the GPU job itself produced, and whether the HIP assignments equal scipy's;
    * fp64 ``torch.linalg.lstsq`` on the accumulated (A, B) of three 3x3 layers (K = 576 / 2304 / 4608) from the closed-form
      leg, scaled to all layers by sum K^3, and how far the HIP Cholesky solution is from it."""
    from scipy.optimize import linear_sum_assignment

    out = {}
    mats = {k: v.detach().cpu().numpy() for k, v in costs.items()}
    t0 = time.time()
    equal = 0
    for k, a in mats.items():
        rows, cols = linear_sum_assignment(a, maximize=True)
        equal += int((torch.from_numpy(cols) == hip_perm[k]).all())
    out["scipy_lap_s"] = round(time.time() - t0, 3)
    out["scipy_lap"] = "scipy.optimize.linear_sum_assignment on the job's %d cost matrices (n up to %d), one thread; " \
                       "HIP assignments identical in %d / %d groups" % (len(mats), max(a.shape[0] for a in mats.values()), equal, len(mats))
    if alt is not None and alt.get("_samples"):
        t_sum = k3_sum = 0.0
        detail = []
        for smp in alt["_samples"]:
            A, Bt = smp["A"].double(), smp["Bt"].double()
            t0 = time.time()
            A.diagonal().add_(smp["ridge"] * float(A.diagonal().mean()))
            W = torch.linalg.lstsq(A, Bt.t()).solution.t()          # (Cout, K), kernel-position-major like W_hip
            dt = time.time() - t0
            rel = float((smp["W_hip"].double() - W).norm() / W.norm().clamp_min(1e-30))
            detail.append("%s K=%d: %.2fs, HIP Cholesky vs lstsq rel-fro %.1e" % (smp["name"], smp["K"], dt, rel))
            t_sum += dt
            k3_sum += float(smp["K"]) ** 3
        total = t_sum / k3_sum * sum(float(k) ** 3 for k in alt["_K"])
        out["lstsq_fp64_s"] = round(total, 2)
        out["lstsq_fp64"] = "torch.linalg.lstsq (fp64) on the accumulated A, B of %s; scaled by sum K^3 over all %d layers" \
                            % ("; ".join(detail), len(alt["_K"]))
    return out


def stale_sources(blob):
    """Files of ``blob["source_sha256"]`` (kernel sources a traffic measurement was taken on, stamped by
    tools/make_traffic_json.py) whose content is no longer what it was; a file without stamps is stale by definition."""
    import hashlib

    stamps = blob.get("source_sha256")
    if not stamps:
        return ["(no source stamp)"]
    bad = []
    for rel, want in stamps.items():
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path) or hashlib.sha256(open(path, "rb").read()).hexdigest() != want:
            bad.append(rel)
    return bad


def gram_flops_per_sample(spec, m1):
    """Sum over tracked nodes of 2*C^2*HW; shapes from a meta trace."""
    from pleas_merging_amd.core.compiler import trace_with_shapes

    gm = trace_with_shapes(m1, ((1, 3, 224, 224),))
    tracked = {ax.key for g in spec.values() for ax in g.node}
    flops = byts = 0
    for node in gm.graph.nodes:
        if node.name in tracked:
            shp = tuple(node.meta["tensor_meta"].shape)
            c = shp[1]
            hw = 1
            for s in shp[2:]:
                hw *= s
            flops += 2 * c * c * hw
            byts += 2 * c * hw * 4
    return flops, byts


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (start N ranks with torch.distributed.run, or run "
                         "`python bench.py --gpus N` without a launcher and it starts them itself)" % (world, args.gpus))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.all_ranks_on_gpu0:
            local = 0
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from pleas_merging_amd import build as hip_build, hip_ops
    from pleas_merging_amd.core.compiler import get_permutation_spec

    if rank == 0:
        hip_build.build()
    if world > 1:
        dist.barrier()

    n_match, n_pleas = max(1, args.match_batches), max(1, args.updates)
    full = (n_match, n_pleas) == (FULL_MATCH, FULL_PLEAS)
    n_sched = max(1, n_pleas - 1)  # CosineAnnealingLR(T_max=MAX_STEPS) with MAX_STEPS + 1 updates
    m1, m2 = build_models(args.arch, device, args.batch)
    log("models built + BN calibrated")
    t0 = time.perf_counter()
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    spec_s = time.perf_counter() - t0
    ranks = max(world, args.emulate_world)
    pool = Pool(max(n_match, n_pleas), args.batch, device, host=(world == 1 and args.emulate_world <= 1) or args.inputs == "host")
    pool.where = args.inputs
    dp = world > 1
    if args.emulate_world > 1:
        assert world == 1, "--emulate-world is a single-process aid"
        os.environ["PLEAS_EMULATE_WORLD"] = str(args.emulate_world)
        os.environ["PLEAS_EMULATE_ALLREDUCE_US"] = str(args.emulate_allreduce_us)
        dp = True
    cfg = {"dp": dp, "ratio": args.ratio, "prefetch_groups": args.prefetch_groups, "prefetch_memory": args.prefetch_memory,
           "shard_optimizer": not args.no_shard_optimizer, "solver": args.solver, "closed_form_per_forward": args.sources_per_forward or None, "gc": args.gc, "match_mode": args.match_mode, "match_per_forward": args.match_per_forward or None, "sources_per_forward": (args.sources_per_forward * ranks) or None, "lookahead": None if args.lookahead < 0 else bool(args.lookahead)}
    log("spec (%d groups, %.2f s on the host, outside `value`) + %d synthetic batches resident" % (len(spec), spec_s, len(pool.items)))

    def job(phases=None):
        return run_job(cfg, spec, m1, m2, pool.loader(0, n_match), pool.loader(0, n_pleas), n_sched, phases)

    # ---- warm-up: W whole jobs on throw-away state (MIOpen find, allocator pools, twin-graph build paths)
    if args.warmup > 0 and dp:
        # torch's MIOpen binding empties the caching allocator whenever it meets a convolution configuration for the
        # first time.  The only source-forward size a job meets LATE is the one left-over update at its end
        # (batch / ranks samples): met there, it would release the pools the job has just filled.  So that size goes first.
        from pleas_merging_amd.methods.pleas_merging import FrozenSources

        early = FrozenSources(m1, m2, data_parallel=True)
        early.launch(pool.items[0])
        torch.cuda.synchronize()
        early.close()
        del early
    for w in range(args.warmup):
        job()
        torch.cuda.synchronize()
        log("warm-up job %d done" % (w + 1))

    # ---- timed region: exactly K jobs.  Events bracket the few-launches-per-step kernels only: bn_act runs ~200 times per
    # update and would pay two event records per launch inside the timed region.
    hip_ops.profile_reset()
    hip_ops.profile_enable(True, skip=() if args.profile_all else ("bn_act",))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    per_job, res, prev_res = [], None, None
    t0 = time.perf_counter()
    for k in range(args.steps):
        tj = time.perf_counter()
        prev_res = res          # a reference only: compared with the last job's result after the timed region
        res = job()
        torch.cuda.synchronize()
        per_job.append(time.perf_counter() - tj)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    hip_ops.profile_enable(False)
    log("timed region: %d jobs in %.3fs (per job: %s; HBM reserved by the caching allocator: peak %.1f GB)"
        % (args.steps, elapsed, " ".join("%.3f" % t for t in per_job), torch.cuda.max_memory_reserved(device) / 1e9))
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = elapsed / max(1, args.steps)
    prof = hip_ops.profile_collect() if rank == 0 else {}  # the timed jobs' kernels, before anything else is timed
    fwd_lanes = hip_ops.fwd_plan_lanes() if rank == 0 else None   # forms of the grouped forward: measured ms, lane
    checks = check_result(spec, res, full) if rank == 0 else None
    if rank == 0 and prev_res is not None and world == 1 and args.emulate_world <= 1:
        # the same job on the same batches twice in a row inside the timed region: the same bits?  (costs, assignments, every
        # tensor of the merged model; tests/test_hip_determinism.py holds the library to it)
        sd_a, sd_b = prev_res["m3"].state_dict(), res["m3"].state_dict()
        checks["last_two_timed_jobs_bit_identical"] = bool(
            all(torch.equal(prev_res["costs"][k], res["costs"][k]) and torch.equal(prev_res["perm"][k], res["perm"][k]) for k in spec)
            and all(torch.equal(v, sd_b[k]) for k, v in sd_a.items()))
    prev_res = None

    # ---- untimed extras (every rank takes part in the phases job: it contains collectives)
    phases = None
    if not args.no_phases:
        ph = Phases(on=True, verbose=args.phase_log)
        job(ph)
        phases = dict(ph.out)
    alt = vendor = bn_reset = inputs_cmp = None
    if world == 1 and args.emulate_world <= 1 and pool.host_items:
        # the other input mode on extra jobs (one untimed first: its allocator pools / pinned staging are new)
        other = "host" if args.inputs == "resident" else "resident"
        pool.where = other
        job()
        torch.cuda.synchronize()
        t_other = []
        for _ in range(2):
            tj = time.perf_counter()
            job()
            torch.cuda.synchronize()
            t_other.append(time.perf_counter() - tj)
        pool.where = args.inputs
        mb = pool.items[0].numel() * 4 * (n_match + n_pleas) / 1e6
        inputs_cmp = {"timed_jobs_take_batches_from": args.inputs, other + "_s_per_job": round(min(t_other), 4),
                      other + "_jobs": [round(t, 4) for t in t_other],
                      "host_to_device_MB_per_job": round(mb, 1),
                      "note": "host = pinned host tensors, copied inside the loops on a copy stream beside the previous "
                              "batch's compute (hip_ops.to_device_async; reference: blocking x.cuda() per batch, "
                              "activation_matching.py:121, pleas_merging.py:266)"}
        log("inputs from %s: %s s per job (timed jobs, from %s: %.3f)" % (other, inputs_cmp[other + "_s_per_job"], args.inputs, value))
    alt_emulated = None
    if world == 1 and args.emulate_world > 1 and not args.no_alt_solver:
        # rank 0's share of the CLOSED FORM under data parallelism: whole batches b % W == 0 are accumulated, ONE exchange of
        # the A / B arenas at the end (skipped here), every rank solves every layer
        mine = pool.loader(0, n_pleas)[::args.emulate_world]
        a = time_normal_eq(spec, m1, m2, res["perm"], res["costs"], mine, args.ratio, args.sources_per_forward or None)
        alt_emulated = {k: v for k, v in a.items() if not k.startswith("_")}
        alt_emulated["note"] = ("rank 0's share (%d of %d batches) of solver='normal_eq' in a %d-rank job; the all-reduce of the "
                                "A / B arenas before the solve is NOT included" % (len(mine), n_pleas, args.emulate_world))
        log("closed form, emulated rank of %d: accumulate %.2fs, solve %.2fs" % (args.emulate_world, a["accumulate_s"], a["solve_s"]))
    if world == 1 and args.emulate_world <= 1:
        if not args.no_alt_solver and args.solver == "adam":
            alt = time_normal_eq(spec, m1, m2, res["perm"], res["costs"], pool.loader(0, n_pleas), args.ratio,
                                 cfg["sources_per_forward"])
            log("closed form: accumulate %.2fs, solve %.2fs" % (alt["accumulate_s"], alt["solve_s"]))
        bn_reset = time_bn_reset(res["m3"], pool)
        log("BN reset of the merged model: HIP path %.2fs, vendor modules %.2fs" % (bn_reset["hip_s"], bn_reset["vendor_modules_s"]))
        src_s, src_per = time_sources_alone(m1, m2, pool, n_pleas, per=cfg["sources_per_forward"] or 2)
        vendor = {"source_forwards_alone_s_per_job": round(src_s, 3), "ms_per_update": round(src_per * 1e3, 3),
                  "share_of_value": round(src_s / value, 3),
                  "note": "frozen source forwards of the PLeaS phase (convolutions as config.source_convolutions says, "
                          "both models, %d samples per forward) with nothing else on the GPU; the key keeps its rounds 1-4 name; "
                          "rocprofv3 kernel shares: profiles/" % ((cfg["sources_per_forward"] or 2) * args.batch)}
    alt_arith = None
    if world == 1 and args.emulate_world <= 1 and not args.no_alt_arith and args.solver == "adam":
        alt_arith = time_alt_arith(job, res, spec, value, n_match, world, rank, args)
    library = None
    if world == 1 and args.emulate_world <= 1 and not args.no_library_baseline and args.solver == "adam":
        library = library_baseline(cfg, spec, m1, m2, pool, n_match, n_pleas, n_sched, args.cpu_match_batches, args.cpu_updates,
                                   res["perm"], res["costs"])
        library["own_kernels_speedup"] = round(library["value"] / value, 2)
        gc.collect()
        torch.cuda.empty_cache()
    if rank == 0:  # {kernel: (launches, total_ms, flops, bytes)}
        labels = {
            "gram_partial": "gram_batch_kernel (grouped fp32 MFMA 32x32x2 contraction, one launch per matching batch)",
            "conv_wgrad": "wgrad_batch_kernel (grouped fp32 MFMA 32x32x2 weight gradients, one launch per PLeaS update)",
            "conv_fwd": "fwd_batch_kernel (grouped fp32 MFMA 32x32x2 forward + target + residual + loss, one launch per PLeaS update)",
        }

        def roof(name):
            rec = prof.get(name, (0, 0.0, 0.0, 0.0))
            ach = rec[2] / (rec[1] * 1e-3) / 1e12 if rec[1] > 0 else 0.0
            return {"bound": "mfma", "kernel": labels[name], "achieved": round(ach, 2), "peak": FP32_MATRIX_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": None,
                    "launches": rec[0], "avg_launch_us": round(rec[1] * 1e3 / max(rec[0], 1), 2),
                    "algorithmic_flop_per_launch": round(rec[2] / max(rec[0], 1)), "total_ms": round(rec[1], 2)}

        roofs = {k: roof(k) for k in labels}
        # the plain convolution (k x k layers of the frozen sources / twin forwards): listed, never the dominant kernel -- the two
        # source models' launches run pairwise on two streams, so a launch's HIP-event time includes its sibling's share of the
        # chip (by rocprofv3 kernel time: fwd_batch_kernel 30.7 %, conv2d_fwd_kernel 18.1 %, profiles/r05_a_bench_kernel_stats.csv)
        labels["conv2d"] = "conv2d_fwd_kernel (fp32 MFMA 32x32x2 plain convolution: the k x k layers of the frozen source / twin forwards)"
        conv2d_roof = roof("conv2d")
        conv2d_roof["note"] = ("the two source models' launches share the chip pairwise (two streams): event times include the "
                               "sibling's share; alone at 128 samples the 3 x 3 layers run at 89-94 TFLOP/s, the 1 x 1 layers at "
                               "46-96 (profiles/r05_probe_conv_classes_128.txt)")
        # `achieved` counts the flops the contraction kernel EXECUTES.  The path's algorithmic work per matching batch
        # (6.368e10 flop per sample for ResNet-101, every tracked node contracted) is larger: the 104
        # tracked BatchNorm nodes are derived from their convolution node in the reduce pass, not contracted.
        # one contraction launch covers the batches of one twin forward (--match-per-forward): batches per launch, from the
        # launches this rank made
        gram_batches = 1.0
        if roofs["gram_partial"]["launches"]:
            mine = len(range(rank, n_match, world)) * args.steps
            gram_batches = mine / roofs["gram_partial"]["launches"]
            roofs["gram_partial"]["matching_batches_per_launch"] = round(gram_batches, 3)
        if args.arch == "resnet101" and roofs["gram_partial"]["launches"]:
            per_launch = 6.368e10 * args.batch * gram_batches      # whole batches per rank: matching shards batch indices
            us = roofs["gram_partial"]["avg_launch_us"]
            roofs["gram_partial"]["path_equivalent"] = {
                "flop_per_launch": per_launch, "tflops": round(per_launch / (us * 1e-6) / 1e12, 2) if us else 0.0,
                "note": "all 344 tracked nodes as the reference contracts them; 240 are contracted here, 104 derived"}
        # HBM-side bytes per launch: rocprofv3 PMC passes on standalone replays of the same grids (profiles/*_traffic.json,
        # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); a counter pass cannot run inside this process
        for key, fname in (("gram_partial", "gram_traffic.json"), ("conv_fwd", "fwd_traffic.json"),
                           ("conv_wgrad", "wgrad_traffic.json")):
            tpath = os.path.join(ROOT, "profiles", fname)
            if os.path.exists(tpath) and args.arch == "resnet101" and args.batch == 16 and world == 1:
                blob = json.load(open(tpath))
                stale = stale_sources(blob)
                if stale:       # measured on another version of the kernel's source: not this kernel's traffic
                    roofs[key]["traffic_refused"] = "profiles/%s was measured on another version of %s" % (fname, ", ".join(stale))
                    continue
                roofs[key]["traffic"] = blob.get("hbm_bytes_per_launch")
                if key == "gram_partial" and roofs[key]["traffic"]:      # the file is per matching batch
                    roofs[key]["traffic"] = round(roofs[key]["traffic"] * gram_batches)
                roofs[key]["traffic_source"] = "profiles/" + fname
                if roofs[key]["traffic"] and roofs[key]["avg_launch_us"]:    # north_star: rocprof HBM GB/s of the accumulation
                    roofs[key]["hbm_gbps"] = round(roofs[key]["traffic"] / (roofs[key]["avg_launch_us"] * 1e-6) / 1e9, 1)
        dominant = max(roofs, key=lambda k: roofs[k]["total_ms"])   # own grouped launch with the most time in the timed region
        roofline = roofs[dominant]
        other = {k: v for k, v in roofs.items() if k != dominant}
        if conv2d_roof["launches"]:
            other["conv2d"] = conv2d_roof
        srt = sorted(per_job)
        out = {
            "metric": "wall-clock (s): ResNet-101 pair, 100-batch act-match + 400-step PLeaS, 1/8 GPU"
            if args.arch == "resnet101" else "wall-clock (s): %s pair, act-match + PLeaS" % args.arch,
            "value": round(value, 4), "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(value * 1e3, 2), "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s pair (random init, BN calibrated), ONE JOB per step: %d matching batches + LAP (%d groups) + "
                            "partial merge (ratio %.2f) + %d PLeaS Adam updates, batch %d x 3x224x224%s"
                            % (args.arch, n_match, len(spec), args.ratio, n_pleas, args.batch,
                               "" if full else " (SHORTENED job: not a benchmark result)"),
                "solver": args.solver, "parallelism": "dp%d" % world,
                "sources_per_forward": cfg["sources_per_forward"] or 2 * ranks,
                "matching_batches_per_forward": cfg["match_per_forward"] or 2,
                "host_gc": "collected once per job while the LAP kernel runs" if args.gc == "lap" else "interpreter default",
                "inputs": "HBM-resident batches" if args.inputs == "resident" else "pinned host batches, copied host -> device inside the loops",
                "matching_mode": args.match_mode,
                "source_convolutions": {"kxk": "k x k layers of the frozen source / twin forwards on the library's own kernel "
                                                   "(pleas_conv2d_fwd: repeatable bits), 1 x 1 layers on the vendor's GEMM",
                                            "all": "every convolution of the frozen source / twin forwards on the library's own kernel"
                                                   + (", the PLeaS phase's with their BatchNorm / add / ReLU chain in the epilogue"
                                                      if os.environ.get("PLEAS_SOURCE_CONV_BN", "1") == "1" else ""),
                                            "vendor": "MIOpen / Tensile for every convolution of the frozen source / twin forwards "
                                                      "(rounds 1-4; not run-to-run deterministic)"}[source_conv_mode()],
                "not_in_value": "get_permutation_spec %.2f s (host, once per model)" % spec_s,
            },
            "job_s": {"mean": round(value, 4), "median": round(srt[len(srt) // 2], 4), "min": round(srt[0], 4),
                      "max": round(srt[-1], 4), "note": "rank 0's per-job times; value = bracketed total / steps"},
            "roofline": roofline,
            "roofline_other": other,
        }
        if args.solver == "normal_eq":
            out["alt_solver_line"] = ("every job's PLeaS phase is the CLOSED FORM (normal equations + layer-sharded Cholesky) instead of "
                                      "the 401 Adam updates: not the headline (BASELINE.json's metric is the Adam-faithful loop)")
            out["metric"] = "ALT-SOLVER " + out["metric"]
        if os.environ.get("PLEAS_GRAM_SPLIT_BF16", "0") == "1":      # the study arithmetic of the matching contraction
            out["study"] = ("matching contraction on bf16 MFMA (three-way split of the fp32 operands, DESIGN.md 4.0): "
                            "NOT the headline path, not a benchmark result")
            out["metric"] = "STUDY " + out["metric"]
            out["dtype"] = "f32 operands as 3 x bf16 (matching contraction only)"
        if args.emulate_world > 1:
            out["emulated"] = ("rank 0's share of a %d-rank job, collectives %s: NOT a benchmark result"
                               % (args.emulate_world, "replaced by a %.0f us stall of the update stream" % args.emulate_allreduce_us
                                  if args.emulate_allreduce_us > 0 else "skipped"))
            out["metric"] = "EMULATED " + out["metric"]
        if world == 1 and not args.no_cpu_baseline and args.emulate_world <= 1 and args.solver == "adam":
            out["cpu_baseline"], parity, parity_alt = cpu_baseline_and_parity(cfg, spec, m1, m2, pool, n_match, n_pleas, n_sched,
                                                                              args.cpu_match_batches, args.cpu_updates,
                                                                              alt_arith=alt_arith is not None)
            if alt_arith is not None and parity_alt is not None:
                alt_arith["parity_vs_fp64"] = parity_alt.pop("parity_vs_fp64")
                alt_arith["parity_vs_oracle"] = parity_alt
                alt_arith["ok"] = bool(parity_alt["ok"])
            out["cpu_baseline"]["reference_legs"] = cpu_reference_legs(res["costs"], res["perm"], alt)
            checks["parity_vs_fp64"] = parity.pop("parity_vs_fp64")
            checks["parity_vs_oracle"] = parity
            checks["ok"] = bool(checks["ok"] and parity["ok"])
        if phases is not None:
            phases = {"spec": round(spec_s, 4), **phases,
                      "note": "one extra job with a device synchronisation at every boundary (phases cannot overlap "
                              "there, so they add up to a little more than `value`); spec is outside `value`"}
            out["phases_s"] = phases
        if alt is not None:
            alt = {k: v for k, v in alt.items() if not k.startswith("_")}
            tpath = os.path.join(ROOT, "profiles", "neq_traffic.json")
            if os.path.exists(tpath) and args.arch == "resnet101" and args.batch == 16:
                blob = json.load(open(tpath))
                stale = stale_sources(blob)
                if stale:
                    alt["neq_batch_kernel"]["traffic_refused"] = "profiles/neq_traffic.json was measured on another version of %s" % ", ".join(stale)
                else:
                    alt["neq_batch_kernel"]["traffic"] = blob.get("hbm_bytes_per_launch")
                    alt["neq_batch_kernel"]["traffic_source"] = "profiles/neq_traffic.json"
            if phases is not None:      # the job with the closed form in place of the 401 Adam updates (synchronised phases)
                alt["job_s_if_closed_form"] = round(phases["matching"] + phases["lap"] + phases["merge_and_setup"]
                                                    + alt["accumulate_s"] + alt["solve_s"], 3)
            out["alt_solver"] = alt
        if alt_emulated is not None:
            out["alt_solver_emulated_rank"] = alt_emulated
        if inputs_cmp is not None:
            out["inputs"] = inputs_cmp
        if alt_arith is not None:
            out["alt_arith"] = alt_arith
        if library is not None:
            out["library_baseline"] = library
        if vendor is not None:
            out["vendor"] = vendor
        if bn_reset is not None:
            out["bn_reset"] = bn_reset
        out["checks"] = checks
        out["kernels_ms"] = {k: {"launches": v[0], "total_ms": round(v[1], 2)} for k, v in sorted(prof.items())}
        out["fwd_forms"] = fwd_lanes
        print(json.dumps(out), flush=True)
        if not checks["ok"]:
            log("RESULT CHECK FAILED: %r" % (checks,))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not checks["ok"]:
        sys.exit(3)


if __name__ == "__main__":
    main()
