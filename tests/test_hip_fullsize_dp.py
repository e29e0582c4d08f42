"""BASELINE.json config 4 ("ResNet-101 pair, data-sharded batches, RCCL all-reduce of the per-layer matrices") rehearsed as
far as ONE MI355X allows.

(a) ResNet-101 through the data-parallel path with TWO ranks on the one device (gloo rendezvous: RCCL refuses two ranks
    on one GPU): batch-sharded matching + one all-reduce of the 41 MB cost arena, sample-sharded PLeaS updates with the
    default ``2 * world`` updates per source forward (one full group of 4 and the left-over update), look-ahead, one
    all-reduce of the gradient arena per update -- with and without ``shard_optimizer`` -- against the single-process HIP
    job on the same inputs: identical assignments, weights to 2e-5 (or 3x what the single-process job differs from
    ITSELF by when run twice: the vendor convolutions are not run-to-run deterministic at 101 layers), both ranks
    bit-identical.
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


def _job(path, data_parallel, shard=False, time_collectives=False, calls=None):
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
    m3 = partial_merge(spec, m1, m2, perm, costs, RATIO)
    merged_stem = m3.state_dict()["conv1.weight"].detach().cpu().clone()
    fit = PleasFitter(m1, m2, m3, spec, perm, costs, RATIO, N_UPDATES - 1, data_parallel=data_parallel, shard_optimizer=shard)
    assert list(fit.steps(blob["train"])) == list(range(N_UPDATES))       # default sources_per_forward: 2 * world
    loss = fit.loss_sum.clone()
    job_calls = dict(calls) if calls is not None else None      # the job's own collectives, before any are timed below
    timings = _time_collectives(costs, fit) if time_collectives else None
    info = {"world": fit.world, "shard": fit.shard_optimizer, "fast_updates": fit.fast_updates}
    sd = {k: v.cpu() for k, v in fit.finish().state_dict().items()}
    torch.cuda.synchronize()
    return {"perm": {str(k): v.cpu() for k, v in perm.items()}, "costs": {str(k): v.cpu() for k, v in costs.items()},
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


def _worker(rank, world, port, path, out_path, backend, shard, force):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force:
        os.environ["PLEAS_FORCE_COLLECTIVES"] = "1"
    import torch.distributed as dist

    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    calls = collections.Counter()
    for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor", "all_gather"):
        def counted(*a, _fn=getattr(dist, name), _name=name, **kw):
            calls[_name] += 1
            return _fn(*a, **kw)
        setattr(dist, name, counted)
    try:
        res = _job(path, data_parallel=True, shard=shard, time_collectives=force, calls=calls)
        torch.save(res, out_path % rank)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world, port, path, tmp, backend, shard, force, timeout=600):
    ctx = mp.get_context("spawn")
    out_path = os.path.join(tmp, "res_%s_%s_%%d.pt" % (backend, "shard" if shard else "plain"))
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, out_path, backend, shard, force)) for r in range(world)]
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


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("dp"))
    path = os.path.join(tmp, "inputs.pt")
    _make_inputs(path)
    want = _job(path, data_parallel=False)
    assert want["info"]["world"] == 1
    # The yardstick: the SAME single-process job once more.  The vendor's convolution kernels need not be run-to-run
    # deterministic (split-K GEMMs with atomics at these small batch sizes), and 101 layers amplify that; what two runs of
    # one program differ by is what two partitionings of it can be held to.
    again = _job(path, data_parallel=False)
    spread = {"costs": {k: _rel(again["costs"][k], v) for k, v in want["costs"].items()},
              "sd": {k: _rel(again["sd"][k], v) for k, v in want["sd"].items() if v.dtype.is_floating_point},
              "perm_equal": all(torch.equal(again["perm"][k], v) for k, v in want["perm"].items())}
    print("single-process job twice: worst cost rel-fro %.2e, worst weight rel-fro %.2e, assignments equal: %s"
          % (max(spread["costs"].values()), max(v for k, v in spread["sd"].items() if k != "conv1.weight"), spread["perm_equal"]))
    want["spread"] = spread
    return tmp, path, want


COST_TOL, WEIGHT_TOL = 1e-5, 2e-5      # floors; a tensor may differ by 3x what two runs of the single-process job differ by


def _compare(res, want, exact):
    """``exact``: the run must reproduce ``want`` as well as ``want`` reproduces itself (bit for bit when the
    single-process job is deterministic).  Otherwise rel-fro per tensor within max(floor, 3 x the job's own spread)."""
    spread = want["spread"]
    deterministic = spread["perm_equal"] and max(spread["costs"].values()) == 0.0 and max(spread["sd"].values()) == 0.0
    worst = {"cost": 0.0, "weight": 0.0}
    # A group whose optimum is a near tie may be assigned differently by two partitionings of the job (the all-reduce sums
    # the batches' costs in another order): accepted when the other assignment is within 1e-6 of the optimum under THIS job's
    # costs (the rule of test_hip_fullsize._check_matching), for at most two groups; the tensors that carry such a group's
    # axes are then merged differently and are left out of the weight comparison (layers are fitted independently).
    flipped, skip = [], set()
    value = lambda cost, perm: float(cost.double()[torch.arange(len(perm)), perm].sum())
    for k, v in want["perm"].items():
        if spread["perm_equal"] and not torch.equal(res["perm"][k], v):
            best, mine = value(want["costs"][k], v), value(want["costs"][k], res["perm"][k])
            gap = (best - mine) / abs(best)
            assert not (exact and deterministic) and 0 <= gap < 1e-6, (k, gap)
            flipped.append((k, int((res["perm"][k] != v).sum()), gap))
            skip.update(want["state_axes"][k])
        r = _rel(res["costs"][k], want["costs"][k])
        worst["cost"] = max(worst["cost"], r)
        if exact and deterministic:
            assert torch.equal(res["costs"][k], want["costs"][k]), k
        else:
            assert r <= max(COST_TOL, 3 * spread["costs"][k]), (k, r, spread["costs"][k])
    for k, v in want["sd"].items():
        if not v.dtype.is_floating_point:
            continue
        if exact and deterministic:
            assert torch.equal(res["sd"][k], v), k
        elif k != "conv1.weight" and k not in skip:      # degenerate stem: gated by the caller
            r = _rel(res["sd"][k], v)
            worst["weight"] = max(worst["weight"], r)
            assert r <= max(WEIGHT_TOL, 3 * spread["sd"][k]), (k, r, spread["sd"][k])
    assert len(flipped) <= 2, flipped
    if flipped:
        print("near-tie groups assigned differently (units, optimality gap):", flipped)
    if exact and deterministic:
        assert torch.equal(res["loss"], want["loss"])
    else:
        keep = torch.tensor([not any(a.rsplit(".", 1)[0] == n for a in skip) for n in want["layers"]])
        assert torch.allclose(res["loss"][keep], want["loss"][keep], rtol=1e-4, atol=1e-7)
    return worst


@pytest.mark.parametrize("shard", [False, True])
def test_rn101_two_rank_job_equals_single_process_job(single, shard):
    tmp, path, want = single
    results = _spawn(2, 29671 + int(shard), path, tmp, "gloo", shard, force=False)
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
    tmp, path, want = single
    (res,) = _spawn(1, 29681 + int(shard), path, tmp, "nccl", shard, force=True)
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
