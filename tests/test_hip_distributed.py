"""Two ranks on ONE MI355X (gloo rendezvous, both processes on cuda:0): the data-parallel HIP path end to end --
batch-sharded matching with one all-reduce of the cost arena, sample-sharded PLeaS updates with grouped source
forwards, look-ahead and one all-reduce (gradients + losses) per update -- against the single-process HIP path on the
same inputs.  RCCL refuses two ranks on one device, so the collectives go through gloo here; the calls are the same."""
import copy
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO
from stem_gate import gate_stem, stem_objective

pytestmark = pytest.mark.gpu

N_MATCH, N_UPDATES = 4, 11      # 11 updates: two full groups of 2 * world = 4 batches and a tail of three singles


def _batches(t):
    xs = [x for x, _ in t.batches() + t.batches() + t.batches()]
    return [x + 0.01 * i for i, x in enumerate(xs)][:N_UPDATES]


def _job(t, data_parallel, shard=False, presharded=False):
    from pleas.methods.activation_matching import activation_matching
    from pleas.methods.partial_matching import partial_merge
    from pleas.methods.pleas_merging import PleasFitter

    m1, m2 = copy.deepcopy(t.m1).cuda(), copy.deepcopy(t.m2).cuda()
    if presharded:      # the loader hands every rank ITS batches only (DistributedSampler-style); same partition b % world
        import torch.distributed as dist

        mine = [b for i, b in enumerate(t.batches()[:N_MATCH]) if i % dist.get_world_size() == dist.get_rank()]
        perm, costs = activation_matching(t.spec, m1, m2, mine, len(mine), output_costs=True, presharded=True)
    else:
        perm, costs = activation_matching(t.spec, m1, m2, t.batches(), N_MATCH, output_costs=True)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
    merged_stem = m3.state_dict()["conv1.weight"].detach().cpu().clone()
    fit = PleasFitter(m1, m2, m3, t.spec, perm, costs, 0.5, N_UPDATES - 1, num_classes=10, data_parallel=data_parallel,
                      shard_optimizer=shard)
    assert fit.shard_optimizer == (shard and fit.world > 1) and fit.m.numel() * (fit.world if fit.shard_optimizer else 1) == fit.p.numel()
    assert list(fit.steps(_batches(t))) == list(range(N_UPDATES))
    loss = fit.loss_sum.clone()
    torch.cuda.synchronize()
    sd = {k: v.cpu() for k, v in fit.finish().state_dict().items()}
    sd["__merged_stem"] = merged_stem
    return ({k: v.cpu() for k, v in perm.items()}, {k: v.cpu() for k, v in costs.items()}, sd, loss.cpu(), fit.world)


def _job_normal_eq(t):
    """Matching + closed-form PLeaS (solver="normal_eq", the body of train_normal_eq).  With torch.distributed initialised
    both phases shard WHOLE batches over the ranks: one all-reduce of the cost arena, one of the A / B arenas before the
    solve.  Returns the (summed) arenas as well: most layers of the 4-wide fixture see fewer rows than unknowns, so their
    solved weights are set by the ridge and say little; the arenas are what the exchange step must get right."""
    from pleas.methods.activation_matching import activation_matching
    from pleas.methods.partial_matching import partial_merge
    from pleas_merging_amd.methods.activation_matching import _dist_info
    from pleas_merging_amd.methods.normal_eq import NormalEqFitter

    m1, m2 = copy.deepcopy(t.m1).cuda(), copy.deepcopy(t.m2).cuda()
    perm, costs = activation_matching(t.spec, m1, m2, t.batches(), N_MATCH, output_costs=True)
    m3 = partial_merge(t.spec, m1, m2, perm, costs, 0.5)
    fit = NormalEqFitter(m1, m2, m3, t.spec, perm, costs, 0.5, N_UPDATES - 1, num_classes=10)
    fit.rank, fit.world = _dist_info()
    mine = [x for i, x in enumerate(_batches(t)) if i % fit.world == fit.rank]
    assert len(list(fit.steps(mine))) == len(mine)
    fit.solve()                                  # all-reduces A, B (and the bias statistics) in place, then solves
    arenas = {"A": fit.A_flat.cpu(), "B": fit.B_flat.cpu()}
    sd = {k: v.cpu() for k, v in fit.finish().state_dict().items()}
    torch.cuda.synchronize()
    sd.update({"__" + k: v for k, v in arenas.items()})
    return {k: v.cpu() for k, v in perm.items()}, {k: v.cpu() for k, v in costs.items()}, sd, torch.zeros(1), fit.world


def _worker(rank, world, port, q, buckets):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from conftest import Tiny

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if buckets == "normal_eq":
            perm, costs, sd, loss, world = _job_normal_eq(Tiny("tiny_bottleneck.npz"))
        elif buckets == "shard":
            perm, costs, sd, loss, world = _job(Tiny("tiny_bottleneck.npz"), data_parallel=True, shard=True)
        elif buckets == "presharded":
            perm, costs, sd, loss, world = _job(Tiny("tiny_bottleneck.npz"), data_parallel=True, presharded=True)
        else:
            perm, costs, sd, loss, world = _job(Tiny("tiny_bottleneck.npz"), data_parallel=True)
        as_np = lambda d: {k: v.numpy() for k, v in d.items()}   # plain arrays: nothing shared with a process that exits
        q.put((rank, (as_np(perm), as_np(costs), as_np(sd), loss.numpy(), world)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


# one all-reduce per update / reduce-scatter + sharded Adam + all-gather / the matching loader already split per rank
@pytest.mark.parametrize("buckets", [1, "shard", "presharded"])
def test_two_rank_job_equals_single_process_job(tiny_bottleneck, buckets):
    want_perm, want_costs, want_sd, want_loss, world1 = _job(tiny_bottleneck, data_parallel=False)
    assert world1 == 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29641 + (buckets if isinstance(buckets, int) else 7 + len(buckets))
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, buckets)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in procs)
    as_t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    results = {r: (as_t(p), as_t(c), as_t(sd), torch.from_numpy(l), w) for r, (p, c, sd, l, w) in results.items()}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(results) == [0, 1]
    for rank, (perm, costs, sd, loss, world) in results.items():
        assert world == 2
        for k in want_perm:
            assert torch.equal(perm[k], want_perm[k]), (rank, k)
            assert torch.allclose(costs[k], want_costs[k], rtol=1e-5, atol=1e-5), (rank, k)
        assert torch.allclose(loss, want_loss, rtol=1e-4, atol=1e-7), rank
        for k, v in want_sd.items():
            if not v.dtype.is_floating_point:
                continue
            if k == "conv1.weight":   # degenerate stem: its residual is rounding noise -- travel from
                # the merged value within 1.5x the single-process run's own, and the same layer objective (tests/stem_gate.py)
                t = tiny_bottleneck
                gate_stem(sd[k], want_sd["__merged_stem"], [v],
                          lambda w: stem_objective(t.m1, t.m2, w, t.spec, want_perm, want_costs, 0.5,
                                                   [(x, None) for x in _batches(t)], 10), what="rank %d stem" % rank)
            else:
                assert _rel(sd[k], v) < 2e-5, (rank, k, _rel(sd[k], v))
    # both ranks hold the same model, bit for bit (every rank applies the same all-reduced update)
    for k, v in results[0][2].items():
        assert torch.equal(v, results[1][2][k]), k


def test_two_rank_normal_eq_equals_single_process(tiny_bottleneck):
    """north_star's multi-GPU design for the closed form: batches shard over ranks, ONE all-reduce of the accumulated
    A = U^T U / B = U^T Y arenas, then every rank solves.  Two ranks give the single-process weights (sum order only)."""
    want_perm, want_costs, want_sd, _, world1 = _job_normal_eq(tiny_bottleneck)
    assert world1 == 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29651, q, "normal_eq")) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(results) == [0, 1]
    for rank, (perm, costs, sd, _, world) in results.items():
        assert world == 2
        for k in want_perm:
            assert (torch.from_numpy(perm[k]) == want_perm[k]).all(), (rank, k)
        for k in ("__A", "__B"):      # the all-reduced normal equations: sum order only
            assert _rel(torch.from_numpy(sd[k]), want_sd[k]) < 1e-5, (rank, k, _rel(torch.from_numpy(sd[k]), want_sd[k]))
        # solved weights where the system is well determined (704+ rows per batch set for K <= 72 unknowns: stage 1)
        checked = 0
        for k, v in want_sd.items():
            if k.startswith("layer1.") and k.endswith("weight") and v.dim() == 4:
                assert _rel(torch.from_numpy(sd[k]), v) < 1e-3, (rank, k, _rel(torch.from_numpy(sd[k]), v))
                checked += 1
        assert checked >= 3
    for k, v in results[0][2].items():
        assert (v == results[1][2][k]).all(), k      # both ranks hold the same model


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the way the driver starts --gpus 1): the script starts
    torch.distributed.run itself before touching the GPU, both ranks run the whole data-parallel job (gloo here: RCCL
    refuses two ranks on one device), rank 0 prints ONE valid JSON line and the parent exits with the ranks' status."""
    import json
    import subprocess

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--all-ranks-on-gpu0",
           "--steps", "1", "--warmup", "1", "--arch", "resnet18", "--batch", "4", "--match-batches", "3", "--updates", "9",
           "--prefetch-groups", "1", "--prefetch-memory", "0.05", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["warmup"] == 1 and line["unit"] == "s"
    assert line["config"]["parallelism"] == "dp2" and "SHORTENED" in line["config"]["workload"]
    assert line["value"] > 0 and abs(line["ms_per_step"] - 1e3 * line["value"]) < 1.0
    assert line["checks"]["perms_are_permutations"] and line["checks"]["weights_finite"]
    assert set(line["phases_s"]) >= {"spec", "matching", "lap", "merge_and_setup", "updates"}
    assert line["roofline"]["launches"] > 0 and "cpu_baseline" not in line


def test_bench_two_ranks_closed_form_line():
    """`python bench.py --gpus 2 --solver normal_eq`: the job whose exchange does not grow with the updates (whole batches per
    rank, ONE all-reduce of the A / B arenas, layer-sharded Cholesky, ONE sum of the solved parameter arena) as a supported
    bench line -- tagged ALT-SOLVER, weights finite."""
    import json
    import subprocess

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--all-ranks-on-gpu0",
           "--solver", "normal_eq", "--steps", "1", "--warmup", "1", "--arch", "resnet18", "--batch", "4", "--match-batches", "3",
           "--updates", "9", "--prefetch-groups", "1", "--prefetch-memory", "0.05", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["metric"].startswith("ALT-SOLVER") and line["config"]["solver"] == "normal_eq"
    # (9 batches of 4 leave the 7 x 7 layers with fewer rows than unknowns: those systems take the fp64 fall-back, counted)
    assert line["checks"]["ok"] and line["checks"]["weights_finite"] and line["checks"]["fp64_fallbacks"] >= 0
    assert line["value"] > 0 and "cpu_baseline" not in line and "library_baseline" not in line
