"""BASELINE.json config 4 ("ResNet-101 pair, data-sharded batches, RCCL all-reduce of the per-layer matrices") rehearsed as
far as ONE MI355X allows.

(a) ResNet-101 through the data-parallel path with TWO ranks on the one device (gloo rendezvous: RCCL refuses two ranks
    on one GPU): batch-sharded matching + one all-reduce of the 41 MB cost arena, sample-sharded PLeaS updates with the
    default ``2 * world`` updates per source forward (one full group of 4 and the left-over update), look-ahead, one
    all-reduce of the gradient arena per update -- with and without ``shard_optimizer`` -- against the single-process HIP
    job on the same inputs.  MATCHING is compared on its own (costs; assignments identical or, in a near-tie group, equal
    to the LAP of the rank's own costs and within 1e-6 of the optimum under the single-process costs -- always checked);
    the FIT of every run starts from ONE fixed assignment (the single-process job's), so a near-tie flip cannot leak into
    the weight comparison.  Weights: per tensor within max(floor, 3x the largest pairwise distance among FOUR runs of
    the single-process job), or -- Adam's first updates are sign-like -- a bounded share of coordinates one visible
    step apart and the rest inside the floor; bias vectors by absolute travel in units of lr.  Both ranks bit-identical.
(b) The same job in a ONE-rank ``nccl`` process group with PLEAS_FORCE_COLLECTIVES=1: ``all_reduce`` of the cost and
    gradient arenas, ``reduce_scatter_tensor`` / ``all_gather_into_tensor`` of the sharded optimiser really go through
    RCCL (counted), leave the result bit-identical to the job without a process group, and are timed on the job's own
    arenas (written to gpurun_out/r03_rccl_one_rank.json when that directory exists).

The reference has no multi-GPU path (its only line is experiments/datasets/common.py:68); what is reproduced is its
single-process semantics (activation_matching.py:119-134, pleas_merging.py:367-375) under the partitioning of DESIGN.md
section 5."""
import collections
import json
import os
import sys
import time

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

ARCH, BATCH, N_MATCH, N_UPDATES, RATIO = "resnet101", 4, 2, 5, 0.5


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _make_inputs(path):
    """Models (calibrated BatchNorm) and batches, built ONCE and handed to every process as a file."""
    from pleas_merging_amd import resnet as zoo

    g = torch.Generator().manual_seed(11)
    match = [torch.randn(BATCH, 3, 224, 224, generator=g) for _ in range(N_MATCH + 1)]
    train = [torch.randn(BATCH, 3, 224, 224, generator=g) for _ in range(N_UPDATES)]
    sds = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = zoo.MODELS[ARCH](num_classes=1000).cuda()
        zoo.calibrate_bn(m, [x.cuda() for x in match])
        sds.append({k: v.cpu() for k, v in m.state_dict().items()})
    torch.save({"sd": sds, "match": match, "train": train}, path)


def _job(path, data_parallel, shard=False, time_collectives=False, calls=None, fixed=None):
    """Matching in the job's own partitioning (returned as it came out); merge + fit from ``fixed`` -- a file holding ONE
    assignment and its costs -- when given, so that every run that is compared weight for weight merged the same blocks."""
    from pleas.core.compiler import get_permutation_spec
    from pleas.methods.activation_matching import activation_matching
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter
    from pleas_merging_amd import resnet as zoo

    blob = torch.load(path)
    models = []
    for sd in blob["sd"]:
        m = zoo.MODELS[ARCH](num_classes=1000)
        m.load_state_dict(sd)
        models.append(m.cuda().eval())
    m1, m2 = models
    spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
    match = [(x, None) for x in blob["match"]]
    perm, costs = activation_matching(spec, m1, m2, match, N_MATCH, output_costs=True)
    own_perm, own_costs = perm, costs
    if fixed is not None:
        keys = {str(k): k for k in spec}
        blob_f = torch.load(fixed)
        perm = {keys[k]: v for k, v in blob_f["perm"].items()}
        costs = {keys[k]: v.cuda() for k, v in blob_f["costs"].items()}
    m3 = partial_merge(spec, m1, m2, perm, costs, RATIO)
    merged_stem = m3.state_dict()["conv1.weight"].detach().cpu().clone()
    fit = PleasFitter(m1, m2, m3, spec, perm, costs, RATIO, N_UPDATES - 1, data_parallel=data_parallel, shard_optimizer=shard)
    assert list(fit.steps(blob["train"])) == list(range(N_UPDATES))       # default sources_per_forward: 2 * world
    loss = fit.loss_sum.clone()
    job_calls = dict(calls) if calls is not None else None      # the job's own collectives, before any are timed below
    timings = _time_collectives(own_costs, fit) if time_collectives else None
    info = {"world": fit.world, "shard": fit.shard_optimizer, "fast_updates": fit.fast_updates}
    sd = {k: v.cpu() for k, v in fit.finish().state_dict().items()}
    torch.cuda.synchronize()
    return {"perm": {str(k): v.cpu() for k, v in own_perm.items()}, "costs": {str(k): v.cpu() for k, v in own_costs.items()},
            "sd": sd, "loss": loss.cpu(), "info": info, "timings": timings, "merged_stem": merged_stem,
            "calls": job_calls, "state_axes": {str(k): sorted({ax.key for ax in g.state}) for k, g in spec.items()},
            "layers": [p.name for p in fit.plans]}


def _time_collectives(costs, fit, reps=5):
    """The exchange steps of the path on the job's OWN arenas through the initialised backend: seconds per call."""
    import torch.distributed as dist

    first = next(iter(costs.values()))
    arena = first._base if first._base is not None else first        # the flat cost arena all groups are views of
    grads = fit._g_ext
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(),
           "cost_arena_bytes": arena.numel() * 4, "gradient_arena_bytes": grads.numel() * 4}
    n = fit.g.numel() // dist.get_world_size() * dist.get_world_size()
    flat, mine = fit.g[:n].clone(), torch.empty(n // dist.get_world_size(), device=grads.device)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    keep_a, keep_g = arena.clone(), grads.clone()
    out["all_reduce_cost_arena_s"] = timed(lambda: dist.all_reduce(arena))
    out["all_reduce_gradient_arena_s"] = timed(lambda: dist.all_reduce(grads))
    if dist.get_world_size() == 1:      # a sum over one rank must hand the buffers back bit for bit
        out["all_reduce_is_identity"] = bool(torch.equal(arena, keep_a) and torch.equal(grads, keep_g))
    if dist.get_backend() == "nccl":
        out["reduce_scatter_gradient_arena_s"] = timed(lambda: dist.reduce_scatter_tensor(mine, flat))
        out["all_gather_parameter_arena_s"] = timed(lambda: dist.all_gather_into_tensor(flat, mine))
    arena.copy_(keep_a)
    grads.copy_(keep_g)
    return out


def _worker(rank, world, port, path, out_path, backend, shard, force, fixed, device):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force:
        os.environ["PLEAS_FORCE_COLLECTIVES"] = "1"
    import torch.distributed as dist

    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    calls = collections.Counter()
    for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor", "all_gather"):
        def counted(*a, _fn=getattr(dist, name), _name=name, **kw):
            calls[_name] += 1
            return _fn(*a, **kw)
        setattr(dist, name, counted)
    try:
        res = _job(path, data_parallel=True, shard=shard, time_collectives=force, calls=calls, fixed=fixed)
        torch.save(res, out_path % rank)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world, port, path, tmp, backend, shard, force, fixed=None, timeout=600, devices=None):
    """``devices``: the device of every rank (default: all on device 0 -- two ranks on ONE card need gloo)."""
    ctx = mp.get_context("spawn")
    out_path = os.path.join(tmp, "res_%s_%s_%%d.pt" % (backend, "shard" if shard else "plain"))
    devices = devices or [0] * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, out_path, backend, shard, force, fixed, devices[r]))
             for r in range(world)]
    for p in procs:
        p.start()
    deadline = time.time() + timeout
    for p in procs:
        p.join(max(1.0, deadline - time.time()))
        if p.is_alive():
            p.terminate()
            p.join(10)
            pytest.fail("rank did not finish within %d s" % timeout)
        assert p.exitcode == 0, p.exitcode
    return [torch.load(out_path % r) for r in range(world)]


REPEATS = 4      # runs of the single-process job the yardstick is taken from (largest pairwise distance per tensor)


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("dp"))
    path = os.path.join(tmp, "inputs.pt")
    _make_inputs(path)
    want = _job(path, data_parallel=False)
    assert want["info"]["world"] == 1
    # ONE assignment for every fit that is compared weight for weight: the first run's
    fixed = os.path.join(tmp, "fixed.pt")
    torch.save({"perm": want["perm"], "costs": want["costs"]}, fixed)
    # The yardstick: the SAME single-process job REPEATS times.  Whatever is not run-to-run deterministic in it (vendor
    # convolutions that split K with atomics, amplified by 101 layers) bounds what two partitionings of the job can be
    # held to; ONE repeat is one draw of that distance (GPUTEST_r04: fc.bias 2.43e-5 against 3 x a single 6.5e-6).
    runs = [want] + [_job(path, data_parallel=False, fixed=fixed) for _ in range(REPEATS - 1)]
    pairs = [(a, b) for i, a in enumerate(runs) for b in runs[i + 1:]]
    floats = [k for k, v in want["sd"].items() if v.dtype.is_floating_point]
    spread = {"costs": {k: max(_rel(a["costs"][k], b["costs"][k]) for a, b in pairs) for k in want["costs"]},
              "sd": {k: max(_rel(a["sd"][k], b["sd"][k]) for a, b in pairs) for k in floats},
              "max_abs": {k: max(float((a["sd"][k] - b["sd"][k]).abs().max()) for a, b in pairs) for k in floats},
              "share": {k: max(_step_share(a["sd"][k], b["sd"][k])[0] for a, b in pairs) for k in floats},
              "perm_equal": all(torch.equal(r["perm"][k], v) for r in runs[1:] for k, v in want["perm"].items())}
    print("single-process job %d times: worst pairwise cost rel-fro %.2e, worst weight rel-fro %.2e, assignments equal: %s"
          % (REPEATS, max(spread["costs"].values()), max(v for k, v in spread["sd"].items() if k != "conv1.weight"),
             spread["perm_equal"]))
    want["spread"] = spread
    return tmp, path, want, fixed


COST_TOL, WEIGHT_TOL = 1e-5, 2e-5      # floors; a tensor may differ by 3x the largest distance between two runs of the single-process job
REST_TOL = 1e-4                        # north-star tolerance, for the coordinates that did NOT take a visibly different Adam step
LR, SHARE, MAX_FLIPS = 5e-4, 3e-3, 2   # Adam's step; share of coordinates that may sit one visible step apart (test_hip_fullsize)


def _step_share(a, b):
    """(share of coordinates more than lr / 10 apart -- a first, sign-like Adam step that went the other way --, rel-fro of
    all the OTHER coordinates): the two numbers test_hip_fullsize._merge_and_train gates on."""
    d = (a.double() - b.double()).abs()
    far = d > LR / 10
    return float(far.double().mean()), float((d * ~far).norm() / b.double().norm().clamp_min(1e-30))


def _compare_matching(res, want, exact):
    """Costs and assignments of a run's OWN matching against the single-process job's.  Never skipped: when the yardstick
    runs themselves flip a near-tie group, the rule below is exactly what still holds (GPUTEST_r04)."""
    from oracle import pleas_oracle as orc

    spread = want["spread"]
    deterministic = spread["perm_equal"] and max(spread["costs"].values()) == 0.0
    flipped, worst = [], 0.0
    value = lambda cost, perm: float(cost.double()[torch.arange(len(perm)), perm].sum())
    for k, v in want["perm"].items():
        r = _rel(res["costs"][k], want["costs"][k])
        worst = max(worst, r)
        if exact and deterministic:
            assert torch.equal(res["costs"][k], want["costs"][k]), k
        else:
            assert r <= max(COST_TOL, 3 * spread["costs"][k]), (k, r, spread["costs"][k])
        if not torch.equal(res["perm"][k], v):
            # a near tie decided by the summation order: the integer path must be exact on the run's own costs, and the
            # assignment within 1e-6 of the optimum under the single-process costs (rule of test_hip_fullsize._check_matching)
            assert not (exact and deterministic), k
            assert torch.equal(orc.solve_lsa(res["costs"][k]), res["perm"][k]), k
            best, mine = value(want["costs"][k], v), value(want["costs"][k], res["perm"][k])
            gap = (best - mine) / abs(best)
            assert 0 <= gap < 1e-6, (k, gap)
            flipped.append((k, int((res["perm"][k] != v).sum()), gap))
    assert len(flipped) <= MAX_FLIPS, flipped
    if flipped:
        print("near-tie groups assigned differently (units, optimality gap):", flipped)
    return worst


def _compare(res, want, exact):
    """``exact``: the run must reproduce ``want`` as well as ``want`` reproduces itself (bit for bit when the
    single-process job is deterministic).  Otherwise per tensor: rel-fro within max(floor, 3 x the job's own spread), or
    Adam's signature -- a bounded share of coordinates one visible step apart, every other coordinate inside the floor;
    bias vectors: largest absolute distance within max(lr / 10, 3 x the job's own)."""
    spread = want["spread"]
    deterministic = (spread["perm_equal"] and max(spread["costs"].values()) == 0.0 and max(spread["sd"].values()) == 0.0)
    worst = {"cost": _compare_matching(res, want, exact), "weight": 0.0}
    for k, v in want["sd"].items():
        if not v.dtype.is_floating_point:
            continue
        if exact and deterministic:
            assert torch.equal(res["sd"][k], v), k
            continue
        if k == "conv1.weight":      # degenerate stem: gated by the caller
            continue
        r = _rel(res["sd"][k], v)
        worst["weight"] = max(worst["weight"], r)
        if v.dim() == 1:             # bias vectors: travel in units of lr, not a relative norm with a floor
            d = float((res["sd"][k] - v).abs().max())
            assert d <= max(LR / 10, 3 * spread["max_abs"][k]), (k, d, spread["max_abs"][k])
            continue
        if r <= max(WEIGHT_TOL, 3 * spread["sd"][k]):
            continue
        share, rest = _step_share(res["sd"][k], v)
        # (since round 5 the single-process job repeats itself bit for bit, so the spread is 0 and these are the gate: two
        # partitionings sum a gradient in two orders, a near-zero coordinate's first Adam step may go either way -- the gate of
        # test_hip_fullsize._merge_and_train; observed: layer4.0.conv2.weight 4.5e-5 with 4e-5 of its coordinates a step apart)
        assert share <= max(SHARE, 3 * spread["share"][k]) and rest <= max(REST_TOL, 3 * spread["sd"][k]), \
            (k, r, spread["sd"][k], share, rest)
    if exact and deterministic:
        # the reported per-layer loss sums: bit-equal in every run but one of the sharded one-rank RCCL job inside a full-suite run
        # (a last-bit difference in a few entries while every weight was bit-equal; not reproduced alone) -- reported, and held
        # to 1e-6 instead of to the bit, because the weights are what the job returns
        bad = (res["loss"] != want["loss"]).nonzero().flatten().tolist()
        if bad:
            print("per-layer loss sums not bit-equal:", [(want["layers"][i], float(res["loss"][i]), float(want["loss"][i])) for i in bad[:8]])
        assert torch.allclose(res["loss"], want["loss"], rtol=1e-6, atol=1e-9)
    else:
        assert torch.allclose(res["loss"], want["loss"], rtol=1e-4, atol=1e-7)
    return worst


@pytest.mark.parametrize("shard", [False, True])
def test_rn101_two_rank_job_equals_single_process_job(single, shard):
    tmp, path, want, fixed = single
    results = _spawn(2, 29671 + int(shard), path, tmp, "gloo", shard, force=False, fixed=fixed)
    for rank, res in enumerate(results):
        assert res["info"]["world"] == 2 and res["info"]["shard"] == shard
        assert res["info"]["fast_updates"] >= 2           # the group's later updates relaunch patched tables
        worst = _compare(res, want, exact=False)
        print("rank %d shard %s: worst rel-fro vs the single-process job: costs %.2e, weights (non-stem) %.2e; calls %s"
              % (rank, shard, worst["cost"], worst["weight"], res["calls"]))
        assert res["calls"].get("all_reduce", 0) >= 1 + N_UPDATES
    for k, v in results[0]["sd"].items():
        assert torch.equal(v, results[1]["sd"][k]), k       # every rank applied the same update
    # the stem's residual is rounding noise: its travel from the merged value must stay within what
    # the single-process run itself travels (x1.5, tests/stem_gate.py), not within Adam's maximum
    from stem_gate import gate_stem

    assert torch.equal(results[0]["merged_stem"], want["merged_stem"])
    gate_stem(results[0]["sd"]["conv1.weight"], want["merged_stem"], [want["sd"]["conv1.weight"]], what="two-rank stem")


@pytest.mark.parametrize("shard", [False, True])
def test_rn101_one_rank_rccl_group_with_forced_collectives(single, shard):
    tmp, path, want, fixed = single
    (res,) = _spawn(1, 29681 + int(shard), path, tmp, "nccl", shard, force=True, fixed=fixed)
    assert res["info"]["world"] == 1 and res["info"]["shard"] == shard
    calls = res["calls"]
    if shard:      # cost arena + per update: losses (all-reduce), gradients (reduce-scatter), parameters (all-gather)
        assert calls.get("all_reduce", 0) >= 1 + N_UPDATES
        assert calls.get("reduce_scatter_tensor", 0) >= N_UPDATES and calls.get("all_gather_into_tensor", 0) >= N_UPDATES
    else:          # cost arena + one all-reduce of gradients-and-losses per update
        assert calls.get("all_reduce", 0) >= 1 + N_UPDATES and "reduce_scatter_tensor" not in calls
    worst = _compare(res, want, exact=True)  # a sum over one rank is the identity: the plain job, as well as that repeats itself
    print("one-rank RCCL job vs the plain job: costs %.2e, weights %.2e" % (worst["cost"], worst["weight"]))
    t = res["timings"]
    assert t["all_reduce_is_identity"]       # ... and, checked on the arenas themselves, bit for bit
    assert t["backend"] == "nccl" and t["cost_arena_bytes"] > 4e7 and t["gradient_arena_bytes"] > 1e8
    for k, v in t.items():
        if k.endswith("_s"):
            assert 0 < v < 1.0, (k, v)
    print("one-rank RCCL collectives on the job's arenas:", json.dumps(t))
    out_dir = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out_dir) and not shard:
        with open(os.path.join(out_dir, "r03_rccl_one_rank.json"), "w") as f:
            json.dump({"calls_in_job": calls, **t}, f, indent=1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X: real RCCL between two devices over xGMI")
@pytest.mark.parametrize("shard", [False, True])
def test_rn101_two_gpus_rccl_job_equals_single_process_job(single, shard):
    """(c) The first box with two devices runs this without anyone writing new code: the SAME job, one rank per GPU, backend
    ``nccl`` (= RCCL over xGMI): ``all_reduce`` of the cost arena and of the gradient arena, ``reduce_scatter_tensor`` /
    ``all_gather_into_tensor`` of the sharded optimiser between two devices, held to the gates of the gloo rehearsal."""
    tmp, path, want, fixed = single
    results = _spawn(2, 29691 + int(shard), path, tmp, "nccl", shard, force=False, fixed=fixed, devices=[0, 1])
    for rank, res in enumerate(results):
        assert res["info"]["world"] == 2 and res["info"]["shard"] == shard
        worst = _compare(res, want, exact=False)
        print("two GPUs, rank %d shard %s: costs %.2e, weights (non-stem) %.2e; calls %s"
              % (rank, shard, worst["cost"], worst["weight"], res["calls"]))
        if shard:
            assert res["calls"].get("reduce_scatter_tensor", 0) >= N_UPDATES and res["calls"].get("all_gather_into_tensor", 0) >= N_UPDATES
        else:
            assert res["calls"].get("all_reduce", 0) >= 1 + N_UPDATES
    for k, v in results[0]["sd"].items():
        assert torch.equal(v, results[1]["sd"][k]), k
