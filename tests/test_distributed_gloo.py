"""world_size-2 tests on CPU (gloo) of the two exchange steps: batch-sharded cost accumulation
(one all-reduce of the cost matrices) and sample-sharded PLeaS gradients (one all-reduce of the
flat gradient arena).  The collectives are the same calls the RCCL path makes."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _init(rank, world, port):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _matching_worker(rank, world, port, q):
    _init(rank, world, port)
    from conftest import Tiny
    from oracle import pleas_oracle as orc
    from pleas.core.solvers import scipy_solve_lsa
    from pleas.methods.activation_matching import activation_matching, shard_batches

    t = Tiny("tiny_basic.npz")
    mine = [b for b, _ in enumerate(shard_batches(t.batches(), 4, rank, world))]
    perm, costs = activation_matching(t.spec, t.m1, t.m2, t.batches(), 4, cross_features=orc.cross_features_cdist,
                                      lsa_solver=scipy_solve_lsa, output_costs=True, accumulate=True)
    want_p, want_c = orc.activation_matching(t.spec, t.m1, t.m2, t.batches(), 4, accumulate=True)
    ok = len(mine) == 2
    for k in t.spec:
        ok &= bool(torch.allclose(costs[k], want_c[k], rtol=1e-5, atol=1e-4))
        ok &= bool((perm[k] == want_p[k]).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def _gradient_worker(rank, world, port, q):
    _init(rank, world, port)
    import torch.nn.functional as F

    from pleas.methods.pleas_merging import dp_slice, dp_sum_

    g = torch.Generator().manual_seed(0)
    x, target = torch.randn(8, 6, 9, 9, generator=g), torch.randn(8, 5, 9, 9, generator=g)
    w = torch.randn(5, 6, 3, 3, generator=g, requires_grad=True)
    ((F.conv2d(x, w, padding=1) - target) ** 2).mean().backward()  # full-batch gradient of the PLeaS objective
    xs, ts = dp_slice(x, rank, world), dp_slice(target, rank, world)
    out = F.conv2d(xs, w.detach(), padding=1)
    resid = 2.0 * (out - ts) / (out.numel() * world)
    gw = torch.ops.aten.convolution_backward(resid, xs, w.detach(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                             [False, True, False])[1]
    flat = gw.reshape(-1).clone()
    dp_sum_(flat, world)
    q.put((rank, bool(torch.allclose(flat.view_as(w), w.grad, rtol=1e-5, atol=1e-7))))
    dist.destroy_process_group()


def _bn_reset_worker(rank, world, port, q):
    _init(rank, world, port)
    import copy

    from conftest import Tiny
    from pleas.methods.extras import reset_bn_stats

    t = Tiny("tiny_bottleneck.npz")
    data = t.batches() + t.batches("x")[:3]          # 7 batches -> ranks get 4 and 3
    data = [(x + 0.1 * i, y) for i, (x, y) in enumerate(data)]
    seq = reset_bn_stats(copy.deepcopy(t.m1), data, 6, shard=False)       # the drivers' sequential procedure
    par = reset_bn_stats(copy.deepcopy(t.m1), data, 6, shard=True)        # batches b % 2 == rank, one all-reduce
    ok = True
    for (k, a), (_, b) in zip(seq.state_dict().items(), par.state_dict().items()):
        if "running" in k:
            ok &= bool(torch.allclose(a, b, rtol=1e-5, atol=1e-6))
        elif "num_batches_tracked" in k:
            ok &= int(a) == int(b) == 6
    # cumulative-average BatchNorm (momentum=None): the running statistics are the plain mean over the batches
    cum = copy.deepcopy(t.m1)
    for m in cum.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = None
    seq = reset_bn_stats(copy.deepcopy(cum), data, 6, shard=False)
    par = reset_bn_stats(copy.deepcopy(cum), data, 6, shard=True)
    for (k, a), (_, b) in zip(seq.state_dict().items(), par.state_dict().items()):
        if "running" in k:
            ok &= bool(torch.allclose(a, b, rtol=1e-5, atol=1e-6))
        elif "num_batches_tracked" in k:
            ok &= int(a) == int(b) == 6
    ok &= all(m.momentum is None for m in par.modules() if isinstance(m, torch.nn.BatchNorm2d))
    q.put((rank, ok))
    dist.destroy_process_group()


def _sharded_solve_worker(rank, world, port, q):
    """The exchange of the layer-sharded closed-form solve (methods/normal_eq.py: solve_owners + exchange_solved_): every rank
    solves ITS layers only (here with a CPU solver on identical A, B), zeroes the others' and sums the parameter arena once;
    all ranks must end with the single-process solution bit for bit."""
    _init(rank, world, port)
    from pleas_merging_amd.methods.normal_eq import exchange_solved_, solve_owners

    g = torch.Generator().manual_seed(3)
    K, cout = [7, 40, 12, 40, 3, 25, 18], [5, 9, 4, 16, 8, 2, 6]
    owner = solve_owners(K, cout, world)
    systems = []
    for k, co in zip(K, cout):
        U = torch.randn(4 * k, k, generator=g)
        systems.append((U.t() @ U, torch.randn(co, k, generator=g)))
    sizes = [co * k + co for k, co in zip(K, cout)]            # weight + bias per layer
    full, mine = torch.zeros(sum(sizes)), torch.zeros(sum(sizes))
    views = []
    off = 0
    for (A, Bt), k, co in zip(systems, K, cout):
        views.append((off, co * k, co))
        off += co * k + co
    solve = lambda A, Bt: torch.linalg.solve(A, Bt.t()).t().contiguous()
    layer_views = []
    for i, ((A, Bt), (o, nw, nb)) in enumerate(zip(systems, views)):
        W = solve(A, Bt)
        full[o:o + nw] = W.flatten()
        full[o + nw:o + nw + nb] = float(i + 1)
        w_view, b_view = mine[o:o + nw], mine[o + nw:o + nw + nb]
        w_view.fill_(-7.0)                                      # stale values a non-owner holds before the exchange
        b_view.fill_(-7.0)
        if owner[i] == rank:
            w_view.copy_(W.flatten())
            b_view.fill_(float(i + 1))
        layer_views.append([w_view, b_view])
    exchange_solved_(mine, layer_views, owner, rank, world)
    ok = torch.equal(mine, full) and sorted(set(owner)) == list(range(world))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _run(worker, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, True), (1, True)], results


def test_sharded_cost_accumulation_gloo():
    _run(_matching_worker, 29611)


def test_sharded_pleas_gradient_gloo():
    _run(_gradient_worker, 29612)


def test_sharded_bn_reset_equals_sequential_gloo():
    _run(_bn_reset_worker, 29613)


def test_layer_sharded_solve_exchange_gloo():
    _run(_sharded_solve_worker, 29614)


def test_solve_owners_balance_and_determinism():
    from pleas_merging_amd.methods.normal_eq import solve_owners

    K = [147] + [64, 576, 64] * 3 + [256, 1152, 128] * 4 + [512, 2304, 256] * 23 + [1024, 4608, 512] * 3 + [2048]
    cout = [64] + [64, 64, 256] * 3 + [128, 128, 512] * 4 + [256, 256, 1024] * 23 + [512, 512, 2048] * 3 + [1000]
    assert solve_owners(K, cout, 1) == [0] * len(K)
    for world in (2, 4, 8):
        owner = solve_owners(K, cout, world)
        assert owner == solve_owners(list(K), list(cout), world) and set(owner) == set(range(world))
        load = [0.0] * world
        for k, co, r in zip(K, cout, owner):
            load[r] += k ** 3 / 3.0 + 2.0 * co * k * k
        assert max(load) <= 1.25 * (sum(load) / world), (world, load)      # ResNet-101's layers at 8 ranks: within a quarter of even


def test_dp_slice_rejects_ragged_batches():
    from pleas.methods.pleas_merging import dp_slice

    with pytest.raises(RuntimeError):
        dp_slice(torch.zeros(5, 3), 0, 2)
    assert dp_slice(torch.arange(8).view(8, 1), 1, 2).flatten().tolist() == [4, 5, 6, 7]
