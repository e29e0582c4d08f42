"""Closed-form PLeaS: per-layer normal equations accumulated on the GPU, then one solve per layer.

``train(..., solver="normal_eq")``.  The reference fits every merged Conv2d/Linear layer L by Adam on
``mean((L(ip) - op)^2)`` (pleas/methods/pleas_merging.py:281-291, :357-358); the minimiser of the
same objective summed over the batches is ``W^T = A^-1 B`` with ``U = im2col(ip)``,
``A = sum U^T U`` and ``B = sum U^T op``.  (With equal batch sizes the per-batch ``mean`` only
rescales the objective.)  The stacked objectives (``merging`` = reg_mean / perm_separatels / perm_mixedls, :125-145) sum both
half-batches into the same ``A`` and ``B``.  Frozen entries of the reference's gradient mask (:57-58) stay at their
initial value and move to the right-hand side.

MI355X design: per batch, ONE grouped fp32-MFMA launch accumulates ``U^T U`` of every layer
(``pleas_normal_eq_accum``: lower-triangular block tiles, im2col never materialised) and ONE adds
``op . U`` (``pleas_wgrad_batch`` with ACCUMULATE | KPOS_MAJOR).  All ``A`` (and all ``B``) live in one
flat fp32 arena, so a multi-GPU run shards whole batches over ranks and all-reduces each arena ONCE
before the solve.  The K x K systems of all layers (and mask patterns) are solved by ONE batched call of
the HIP blocked Cholesky (``pleas_cholesky_solve_batched``); a system whose fp32 factorisation hits a
non-positive pivot is redone in fp64 on the vendor solver and counted in ``info["fp64_fallbacks"]``.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from .pleas_merging import PleasFitter, _LayerPlan, dp_sum_


class NormalEqFitter(PleasFitter):
    """Accumulates A and B^T for every merged layer; ``solve`` writes the closed-form weights."""

    def __init__(self, *args, ridge: float = 1e-6, shard_solve: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        if self.vendor_layers:      # the Adam-faithful fitter fits such layers on the vendor's operators; the closed form does not
            raise NotImplementedError("solver='normal_eq' takes dense, undilated Conv2d layers with a square kernel / stride / "
                                      "padding; not supported: %s (solver='adam' fits them)" % ", ".join(self.vendor_layers))
        self.ridge = ridge
        # data parallel: rank r factorises only ITS layers (dealt by K^3, largest first) and the solved parameter arena is
        # summed once over the ranks (the others' layers are zero there): the solve shrinks with the ranks instead of being
        # repeated on every one (0.2 s of an 8-rank job's 1.6 s)
        self.shard_solve = bool(shard_solve)
        dev = self.device
        self.K: List[int] = []
        for plan in self.plans:
            w = plan.w_shape
            self.K.append(int(w[1] * (w[2] * w[3] if len(w) == 4 else 1)))
        self.A_flat = torch.zeros(sum(k * k for k in self.K), dtype=torch.float32, device=dev)
        self.B_flat = torch.zeros(sum(k * p.w_shape[0] for k, p in zip(self.K, self.plans)), dtype=torch.float32, device=dev)
        self.A: List[torch.Tensor] = []
        self.Bt: List[torch.Tensor] = []   # [Cout][K] with kernel-position-major columns
        oa = ob = 0
        for k, plan in zip(self.K, self.plans):
            self.A.append(self.A_flat[oa:oa + k * k].view(k, k))
            oa += k * k
            co = plan.w_shape[0]
            self.Bt.append(self.B_flat[ob:ob + co * k].view(co, k))
            ob += co * k
        # bias columns (Linear layers): column sums of U, of op, and the row count
        self.bias_stats: Dict[int, List[torch.Tensor]] = {
            i: [torch.zeros(self.K[i], device=dev), torch.zeros(p.w_shape[0], device=dev), torch.zeros(1, device=dev)]
            for i, p in enumerate(self.plans) if p.b is not None}
        self.neq = self.ops.NormalEqBatch(dev)
        self.batches_seen = 0
        self._slice_batch = False   # data parallel here = whole batches per rank (train_normal_eq), never sample slices

    def _hip_geometry_ok(self, plan: _LayerPlan, ip: torch.Tensor) -> bool:
        mod = plan.mod
        if not plan.is_conv:
            return ip.dim() == 2
        return (mod.groups == 1 and mod.dilation == (1, 1) and mod.stride[0] == mod.stride[1]
                and mod.padding[0] == mod.padding[1] and ip.shape[1] >= 16)

    def _step(self, x: torch.Tensor, next_x=None) -> None:
        """Accumulate one batch (no parameter update); driven by :meth:`PleasFitter.step`."""
        ops = self.ops
        self._begin_update(x, next_x)
        KP = ops.WgradBatch.ACCUMULATE | ops.WgradBatch.KPOS_MAJOR
        # The stacked objectives (reg_mean / perm_separatels / perm_mixedls, reference :125-145) put TWO half-batches under
        # the same weights: A and B are the sums over both.  Each half is its own round of grouped launches (two entries of
        # one launch may not accumulate into the same A).
        for h in range(len(self.plans[0].halves) if self.plans else 0):
            for idx, plan in enumerate(self.plans):
                name = plan.name
                if name not in self.t1_in or name not in self.t2_in:
                    if h == 0:
                        print("Key error on %s" % name)
                    continue
                mod = plan.mod
                in_maps, out_maps = plan.halves[h]
                grouped = self._hip_geometry_ok(plan, self.t1_in[name]) and plan.b is None
                merge = self.merge.add if grouped else ops.merge_blocks     # grouped: filled by merge.flush() below
                ip = merge(self.t1_in[name], self.t2_in[name], 1, *in_maps)
                op = merge(self.t1_out[name], self.t2_out[name], 1, *out_maps)
                if self._hip_geometry_ok(plan, ip):
                    if plan.is_conv:
                        geo = (tuple(mod.kernel_size), mod.stride[0], mod.padding[0])
                        self.neq.add(ip, self.A[idx], *geo)
                        self.wgrad.add(op, ip, self.Bt[idx], *geo, flags=KP)
                    else:
                        self.neq.add(ip, self.A[idx])
                        self.wgrad.add(op, ip, self.Bt[idx], flags=KP)
                else:  # stem (3 input channels) / exotic geometry: tiny K, vendor GEMM on an explicit im2col
                    U = F.unfold(ip, mod.kernel_size, mod.dilation, mod.padding, mod.stride)      # B, (ci,kh,kw), L
                    cin, r = ip.shape[1], mod.kernel_size[0] * mod.kernel_size[1]
                    U = U.view(U.shape[0], cin, r, -1).permute(0, 3, 2, 1).reshape(-1, r * cin)    # rows x (r, ci)
                    Y = op.reshape(op.shape[0], op.shape[1], -1).permute(0, 2, 1).reshape(-1, op.shape[1])
                    self.A[idx].addmm_(U.t(), U)
                    self.Bt[idx].addmm_(Y.t(), U)
                if plan.b is not None:
                    # bias = one more column of ones in U: its normal-equation entries are the column sums of U, the
                    # column sums of the target and the row count
                    s = self.bias_stats[idx]
                    if plan.is_conv:
                        kh, kw = mod.kernel_size
                        st, pd = mod.stride[0], mod.padding[0]
                        ho, wo = op.shape[2], op.shape[3]
                        ipp = F.pad(ip, (pd, pd, pd, pd)) if pd else ip
                        # column sums of im2col(ip) in kernel-position-major order k = (kh * KW + kw) * Cin + ci
                        cols = [ipp[:, :, a:a + st * (ho - 1) + 1:st, b:b + st * (wo - 1) + 1:st].sum((0, 2, 3))
                                for a in range(kh) for b in range(kw)]
                        s[0].add_(torch.cat(cols))
                        s[1].add_(op.sum((0, 2, 3)))
                        s[2].add_(float(op.shape[0] * ho * wo))
                    else:
                        rows_u = ip.reshape(-1, ip.shape[-1])
                        s[0].add_(rows_u.sum(0))
                        s[1].add_(op.reshape(-1, op.shape[-1]).sum(0))
                        s[2].add_(float(rows_u.shape[0]))
            self.merge.flush()
            self.neq.flush()
            self.wgrad.flush()
        self.batches_seen += 1
        self._end_update()

    @torch.no_grad()
    def solve(self) -> Dict[str, float]:
        """Closed-form weights for every layer -> the parameter arena.  Returns per-layer info."""
        dp_sum_(self.A_flat, self.world)
        dp_sum_(self.B_flat, self.world)
        for s in self.bias_stats.values():
            for t in s:
                dp_sum_(t, self.world)
        # blocks of stride-1 k x k layers that are copies of contracted ones (lag classes): EVERY matrix of the all-reduced
        # arena, whichever rank contracted into it -- a rank whose share of the batches was empty has seen no geometry of its
        # own and takes the others' (the layers' image sizes; the batch size plays no part)
        mine = self.neq.seen()
        geos = {i: mine[A.data_ptr()][1] for i, A in enumerate(self.A) if A.data_ptr() in mine}
        if self.world > 1:
            import torch.distributed as dist

            if dist.is_initialized():
                every = [None] * dist.get_world_size()
                dist.all_gather_object(every, geos)
                for other in every:
                    for i, g in other.items():
                        geos.setdefault(i, g)
        self.neq.finalize([(self.A[i], g) for i, g in sorted(geos.items())])
        info: Dict[str, float] = {}
        jobs, finals = [], []   # (W, rows, free, A_FF, rhs, A_FF backup) per solve; (plan, W, layout) per layer
        owner = solve_owners(self.K, [p.w_shape[0] for p in self.plans], self.world if self.shard_solve else 1)
        for idx, plan in enumerate(self.plans):
            if owner[idx] != self.rank % max(1, self.world):
                info[plan.name] = float(self.K[idx])
                continue
            K, co = self.K[idx], plan.w_shape[0]
            A = self.A[idx]
            A = torch.tril(A) + torch.tril(A, -1).t()          # kernels fill the lower triangle only
            Bt = self.Bt[idx]
            if plan.kpos:                                      # the arena already holds [Cout][KH][KW][Cin]
                to_kpos = lambda t, co=co, K=K: t.reshape(co, K)
                from_kpos = lambda t, shp=tuple(plan.w.shape): t.reshape(shp)
            elif len(plan.w_shape) == 4:
                cin, r = plan.w_shape[1], plan.w_shape[2] * plan.w_shape[3]
                to_kpos = lambda t, co=co, cin=cin, r=r, K=K: t.reshape(co, cin, r).permute(0, 2, 1).reshape(co, K)
                from_kpos = lambda t, co=co, cin=cin, r=r, shp=plan.w_shape: \
                    t.reshape(co, r, cin).permute(0, 2, 1).reshape(shp)
            else:
                to_kpos = lambda t, co=co, K=K: t.reshape(co, K)
                from_kpos = lambda t, shp=plan.w_shape: t.reshape(shp)
            n_w = plan.w.numel()
            mask = to_kpos(self.mask[plan.off_w:plan.off_w + n_w])
            w0 = to_kpos(plan.w)
            if plan.b is not None:                             # augmented system for the bias column
                su, sy, m = self.bias_stats[idx]
                A = torch.cat([torch.cat([A, su[:, None]], 1), torch.cat([su[None, :], m.view(1, 1)], 1)], 0)
                Bt = torch.cat([Bt, sy[:, None]], 1)
                mask = torch.cat([mask, torch.ones(co, 1, device=mask.device)], 1)
                w0 = torch.cat([w0, plan.b[:, None]], 1)
            W = w0.clone()
            patterns, inverse = _row_patterns(mask > 0.5)
            for p in range(patterns.shape[0]):
                rows = (inverse == p).nonzero().flatten()
                free = patterns[p].nonzero().flatten()
                if free.numel() == 0:
                    continue
                frozen = (~patterns[p]).nonzero().flatten()
                whole = frozen.numel() == 0
                Aff = A.clone() if whole else A.index_select(0, free).index_select(1, free).contiguous()
                rhs = Bt.index_select(0, rows) if whole else Bt.index_select(0, rows).index_select(1, free)
                if frozen.numel():
                    Afz = A.index_select(0, free).index_select(1, frozen)
                    rhs = rhs - w0.index_select(0, rows).index_select(1, frozen) @ Afz.t()
                jobs.append((W, rows, free, Aff, rhs.contiguous(), Aff.clone()))  # last: untouched copy for a fallback
            finals.append((plan, W, from_kpos))
            info[plan.name] = float(K)
        # ONE batched HIP launch sequence solves every (layer, mask pattern) system
        rhs_orig = [j[4].clone() for j in jobs]
        flags = self.ops.cholesky_solve_batched([j[3] for j in jobs], [j[4] for j in jobs], ridge=self.ridge).cpu()
        for (W, rows, free, _, sol, A_orig), rhs0, bad in zip(jobs, rhs_orig, flags.tolist()):
            if bad:  # fp32 Cholesky broke down (ill-conditioned A): redo this one system in fp64 on the vendor solver
                info["fp64_fallbacks"] = info.get("fp64_fallbacks", 0.0) + 1.0
                sol = _spd_solve_fp64(A_orig, rhs0, self.ridge)
            W[rows[:, None], free[None, :]] = sol
        for plan, W, from_kpos in finals:
            if plan.b is not None:
                plan.b.copy_(W[:, -1])
                W = W[:, :-1]
            plan.w.copy_(from_kpos(W.contiguous()))
        if self.shard_solve and self.world > 1:
            exchange_solved_(self.p, [[t for t in (plan.w, plan.b) if t is not None] for plan in self.plans], owner, self.rank,
                             self.world)
            info["solve_shards"] = float(self.world)
        return info


def exchange_solved_(arena: torch.Tensor, layer_views: List[List[torch.Tensor]], owner: List[int], rank: int, world: int) -> None:
    """After a layer-sharded solve: every layer's parameters (views of ``arena``) were written by exactly one rank.  The
    others' layers are zeroed here and the arena is summed ONCE over the ranks -- x + 0 is exact, so every rank ends with
    the owner's bits.  Without peers (PLEAS_EMULATE_WORLD: a projection, not a result) nothing is zeroed."""
    import torch.distributed as dist

    if world <= 1 or not dist.is_initialized():
        return
    for views, r in zip(layer_views, owner):
        if r != rank:
            for t in views:
                t.zero_()
    dp_sum_(arena, world)


def solve_owners(K: List[int], cout: List[int], world: int) -> List[int]:
    """Which rank factorises which layer: longest-processing-time dealing on the solve's flops (K^3 / 3 + 2 Cout K^2), ties by
    index -- a pure function of the layer list, so every rank computes the same table."""
    if world <= 1:
        return [0] * len(K)
    cost = [k ** 3 / 3.0 + 2.0 * co * k * k for k, co in zip(K, cout)]
    load, owner = [0.0] * world, [0] * len(K)
    for idx in sorted(range(len(K)), key=lambda i: (-cost[i], i)):
        r = min(range(world), key=lambda j: (load[j], j))
        owner[idx] = r
        load[r] += cost[idx]
    return owner


def _row_patterns(free: torch.Tensor):
    """Distinct rows of a boolean matrix and, per row, the index of its pattern (``torch.unique(dim=0)`` semantics up to
    the order of the patterns).  ``unique`` over whole rows costs ~12 ms per layer on the GPU; rows are therefore compared
    through two fp64 projections (the gradient mask has at most three distinct row patterns, :57-58)."""
    if bool(free.all()):
        return free[:1], torch.zeros(free.shape[0], dtype=torch.long, device=free.device)
    g = torch.Generator(device="cpu").manual_seed(0x9E3779B9)
    proj = torch.rand(free.shape[1], 2, generator=g, dtype=torch.float64).to(free.device)
    sig = free.double() @ proj
    key = sig[:, 0] * 1.0000001 + sig[:, 1] * 1e3
    _, inverse = torch.unique(key, return_inverse=True)
    n = int(inverse.max()) + 1
    first = torch.full((n,), free.shape[0], dtype=torch.long, device=free.device)
    first.scatter_reduce_(0, inverse, torch.arange(free.shape[0], device=free.device), reduce="amin")
    patterns = free.index_select(0, first)
    if not bool((patterns.index_select(0, inverse) == free).all()):      # projections collided: exact fallback
        return torch.unique(free, dim=0, return_inverse=True)
    return patterns, inverse


def _spd_solve_fp64(A: torch.Tensor, rhs_rows: torch.Tensor, ridge: float) -> torch.Tensor:
    """Rows x of ``x (A + ridge*mean(diag A) I) = rhs_row`` in fp64 (fallback when the fp32 HIP Cholesky flags a
    non-positive pivot)."""
    A64 = A.double()
    A64 = A64 + ridge * A64.diagonal().mean() * torch.eye(A64.shape[0], device=A.device, dtype=torch.float64)
    L, bad = torch.linalg.cholesky_ex(A64)
    if int(bad) != 0:
        return torch.linalg.lstsq(A64, rhs_rows.double().t()).solution.t().float()
    return torch.cholesky_solve(rhs_rows.double().t(), L).t().float()


def train_normal_eq(dataloader, model1, model2, model3, spec, perm, costs, budget_ratios, MAX_STEPS, separate_classifier,
                    num_classes, model_type, verbose, ridge: float = 1e-6, merging: str = "perm_gradmask"):
    """Same data consumption as the Adam loop (``MAX_STEPS + 1`` batches of one pass, reference :368-373);
    with ``torch.distributed`` initialised, rank r accumulates batches ``b % world == r``."""
    from .activation_matching import _dist_info

    fit = NormalEqFitter(model1, model2, model3, spec, perm, costs, budget_ratios, MAX_STEPS, 5e-4, separate_classifier,
                         num_classes, model_type, merging=merging, ridge=ridge)
    fit.rank, fit.world = _dist_info()
    def inputs():
        for idx, batch in enumerate(dataloader):
            if idx > MAX_STEPS:
                break
            if idx % fit.world == fit.rank:
                yield batch[0]

    for n in fit.steps(inputs()):
        if verbose:
            print("normal_eq: accumulated batch %d of this rank" % n)
    fit.solve()
    return fit.finish()
