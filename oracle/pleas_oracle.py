"""ORACLE -- CPU restatement of the reference's matching + merging hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py as the *checker*; never by the product package
(`pleas_merging_amd/`), never the thing shipped or measured as the product.

Every function restates one reference function in plain fp32 PyTorch on the CPU
(the path is floating point: the tolerance-based bar applies) and cites the
reference file:line it follows.  The restatement is *pinned* by the fixtures in
tests/golden/ that were produced by running the reference itself in the build
container (tests/golden/make_golden.py; checked by tests/test_oracle_golden.py).
The LAP is pinned against scipy (the reference's actual solver) as well.

Written for clarity, not speed; no dependency on the product code beyond the
shared ``Axis`` / ``PermutationGroup`` data model.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from copy import deepcopy
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.fx
import torch.nn.functional as F
from torch import nn

from pleas_merging_amd.core.utils import Axis, PermutationSpec, get_attr, set_attr

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB: Optional[ctypes.CDLL] = None


# ============================================================================ LAP
def build_lsap(force: bool = False) -> str:
    """Compile oracle/lsap.c with gcc into oracle/_build/liblsap_oracle.so."""
    os.makedirs(_BUILD, exist_ok=True)
    so = os.path.join(_BUILD, "liblsap_oracle.so")
    src = os.path.join(_HERE, "lsap.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lm"])
    return so


def _lsap_lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_lsap())
        for name, ctype in (("oracle_lsap_f64", ctypes.c_double), ("oracle_lsap_f32", ctypes.c_float)):
            fn = getattr(_LIB, name)
            fn.argtypes = [ctypes.POINTER(ctype), ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
            fn.restype = ctypes.c_int
    return _LIB


def solve_lsa(cost, maximize: bool = True) -> torch.Tensor:
    """Reference: pleas/core/solvers.py:18-33 (``scipy_solve_lsa``): returns ``col_ind``
    (row i of model 1 <-> column col_ind[i] of model 2) as a CPU int64 tensor."""
    a = cost.detach().cpu().numpy() if torch.is_tensor(cost) else np.asarray(cost)
    assert a.ndim == 2 and a.shape[0] == a.shape[1], "square cost expected"
    n = a.shape[0]
    out = np.empty(n, np.int64)
    lib = _lsap_lib()
    if a.dtype == np.float32:
        a = np.ascontiguousarray(a)
        rc = lib.oracle_lsap_f32(a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, int(maximize),
                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    else:
        a = np.ascontiguousarray(a, np.float64)
        rc = lib.oracle_lsap_f64(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, int(maximize),
                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    assert rc == 0, "infeasible cost matrix"
    return torch.from_numpy(out)


def solve_lsa_python(cost: np.ndarray, maximize: bool = True) -> np.ndarray:
    """Pure-Python twin of oracle/lsap.c for small cases (cross-check of the C build)."""
    c = np.asarray(cost, np.float64)
    c = -c if maximize else c.copy()
    n = c.shape[0]
    u, v = np.zeros(n), np.zeros(n)
    col4row, row4col, path = [-1] * n, [-1] * n, [-1] * n
    for cur in range(n):
        remaining = list(range(n - 1, -1, -1))
        shortest = [math.inf] * n
        rows, cols = set(), set()
        dist, i, sink = 0.0, cur, -1
        while sink < 0:
            rows.add(i)
            best, best_at = math.inf, -1
            for t, j in enumerate(remaining):
                r = dist + c[i, j] - u[i] - v[j]
                if r < shortest[j]:
                    shortest[j], path[j] = r, i
                if shortest[j] < best or (shortest[j] == best and row4col[j] < 0):
                    best, best_at = shortest[j], t
            dist = best
            j = remaining[best_at]
            if row4col[j] < 0:
                sink = j
            else:
                i = row4col[j]
            cols.add(j)
            remaining[best_at] = remaining[-1]
            remaining.pop()
        u[cur] += dist
        for r in rows:
            if r != cur:
                u[r] += dist - shortest[col4row[r]]
        for j in cols:
            v[j] -= dist - shortest[j]
        j = sink
        while True:
            r = path[j]
            row4col[j] = r
            col4row[r], j = j, col4row[r]
            if r == cur:
                break
    return np.asarray(col4row, np.int64)


# ============================================================================ cross features
def cross_features_inner_product(x: torch.Tensor, y: torch.Tensor, a: int) -> torch.Tensor:
    """Reference: pleas/methods/activation_matching.py:14-28."""
    xf = torch.movedim(x, a, 0).reshape(x.shape[a], -1)
    yf = torch.movedim(y, a, 0).reshape(y.shape[a], -1)
    return xf @ yf.T


def cross_features_cdist(x: torch.Tensor, y: torch.Tensor, a: int) -> torch.Tensor:
    """Reference: pleas/methods/activation_matching.py:31-46 (negative Euclidean distance)."""
    xf = torch.movedim(x, a, 0).reshape(x.shape[a], -1)
    yf = torch.movedim(y, a, 0).reshape(y.shape[a], -1)
    return -torch.cdist(xf[None], yf[None])[0]


def cross_features_cdist_f64(x: torch.Tensor, y: torch.Tensor, a: int) -> torch.Tensor:
    """fp64 direct-difference distance: the mathematically exact target both fp32 formulas approximate."""
    xf = torch.movedim(x, a, 0).reshape(x.shape[a], -1).double()
    yf = torch.movedim(y, a, 0).reshape(y.shape[a], -1).double()
    return -torch.cdist(xf[None], yf[None], compute_mode="donot_use_mm_for_euclid_dist")[0]


# ============================================================================ activation capture
class _NodeRecorder(torch.fx.Interpreter):
    """Runs a traced model and keeps a copy of each tracked node's value at the moment
    it is produced (i.e. before a later in-place ReLU can overwrite it), which is when
    the reference's cross module evaluates it (activation_matching.py:90-92)."""

    def __init__(self, gm, wanted: Iterable[str]):
        super().__init__(gm)
        self.wanted = set(wanted)
        self.kept: Dict[str, torch.Tensor] = {}

    def run_node(self, n):
        out = super().run_node(n)
        if n.name in self.wanted and torch.is_tensor(out):
            self.kept[n.name] = out.detach().clone()
        return out


@torch.no_grad()
def node_activations(model: nn.Module, x: torch.Tensor, names: Iterable[str]) -> Dict[str, torch.Tensor]:
    gm = torch.fx.symbolic_trace(model)
    rec = _NodeRecorder(gm, names)
    rec.run(x)
    return rec.kept


def matching_costs(spec: PermutationSpec, model1, model2, batches, num_batches: int,
                   cross=cross_features_cdist, accumulate="reference") -> Dict[Axis, torch.Tensor]:
    """Reference: pleas/methods/activation_matching.py:103-136.

    ``accumulate="reference"`` reproduces the shipped behaviour: the membership test at
    :123-127 never succeeds, so every batch overwrites and only the LAST processed batch
    counts.  ``accumulate=True`` is the intended sum over batches.
    Batches are ``(x, label)`` pairs; at most ``num_batches`` are used (zip at :120).
    """
    tracked = [ax for pg in spec.values() for ax in pg.node]
    names = {ax.key for ax in tracked}
    per_node: Dict[Axis, torch.Tensor] = {}
    for (x, _), _ in zip(batches, range(num_batches)):
        a1 = node_activations(model1, x, names)
        a2 = node_activations(model2, x, names)
        for ax in tracked:
            val = cross(a1[ax.key], a2[ax.key], ax.axis)
            if accumulate is True and ax in per_node:
                per_node[ax] = per_node[ax] + val
            else:
                per_node[ax] = val
    costs = {}
    for key, pg in spec.items():
        total = 0
        for nax in pg.node:           # set order, as the reference (:129-134)
            if nax in per_node:
                total = total + per_node[nax]
        costs[key] = total
    return costs


def activation_matching(spec, model1, model2, batches, num_batches=1000, cross=cross_features_cdist,
                        accumulate="reference"):
    """Reference: pleas/methods/activation_matching.py:139-177 -> (perm, costs)."""
    costs = matching_costs(spec, model1, model2, batches, num_batches, cross, accumulate)
    perm = {k: solve_lsa(v) for k, v in costs.items()}
    return perm, costs


# ============================================================================ weight matching
def weight_matching(spec: PermutationSpec, state_a: Dict[str, torch.Tensor], state_b: Dict[str, torch.Tensor],
                    max_iter: int = 100, seed: int = 0, skip_suffixes=("running_mean", "running_var")):
    """Reference: pleas/methods/weight_matching.py:22-95 -> (perm, costs, number_of_LAPs)."""
    state_b = dict(state_b)
    perm = {k: torch.arange(pg.size) for k, pg in spec.items()}
    names = list(perm.keys())
    rng = torch.Generator()
    rng.manual_seed(seed)
    costs, laps = {}, 0
    for _ in range(max_iter):
        progress = False
        for ix in torch.randperm(len(names), generator=rng):
            p = names[ix]
            pg = spec[p]
            A = torch.zeros(pg.size, pg.size)
            for ax in pg.state:
                if ax.key.endswith(tuple(skip_suffixes)) or ax.key not in state_a or ax.key not in state_b:
                    continue
                A += cross_features_inner_product(state_a[ax.key], state_b[ax.key], ax.axis)
            assert A.norm() > 0
            new = solve_lsa(A)
            laps += 1
            old_l, new_l = A.diag().sum(), A[torch.arange(pg.size), new].sum()
            progress = progress or bool(new_l > old_l + 1e-12)
            perm[p] = perm[p][new]
            costs[p] = A
            for ax in pg.state:
                if ax.key in state_b:
                    state_b[ax.key] = torch.index_select(state_b[ax.key], ax.axis, new)
        if not progress:
            break
    return perm, costs, laps


# ============================================================================ partial merge
def get_blocks(spec, perm, costs, ratios) -> Dict[Axis, Tuple[torch.Tensor, ...]]:
    """Reference: pleas/methods/partial_matching.py:47-89.
    Per group -> (merged idx model1, merged idx model2, separate idx model1, separate idx model2)."""
    if not isinstance(ratios, dict):
        ratios = {k: ratios for k in spec}
    blocks = {}
    for key, P in perm.items():
        r = ratios[key]
        if abs(r - 1.0) < 1e-3:
            P = torch.arange(len(P))
        C = costs[key].detach().cpu()
        Q = torch.arange(len(P))
        matched = C[Q, P]
        keep = matched >= torch.quantile(matched, r)
        blocks[key] = (Q[keep], P[keep], Q[~keep], P[~keep])
    return blocks


def spread_blocks(spec, blocks):
    """Every state axis of a group shares the group's blocks (partial_matching.py:100-103)."""
    out = dict(blocks)
    for key, pg in spec.items():
        for ax in pg.state:
            out[ax] = blocks[key]
    return out


def merged_state(spec, sd1, sd2, blocks) -> Dict[str, torch.Tensor]:
    """Reference: pleas/methods/partial_matching.py:91-175 (tensor assembly part)."""
    blocks = spread_blocks(spec, blocks)
    by_tensor: Dict[str, set] = {}
    for pg in spec.values():
        for ax in pg.state:
            by_tensor.setdefault(ax.key, set()).add(ax.axis)
    out = {}
    for name, axes in by_tensor.items():
        if name not in sd1 or name not in sd2:
            continue
        W1, W2 = sd1[name], sd2[name]
        if len(axes) == 1:
            (ax,) = axes
            b1, b2, b1c, b2c = blocks[Axis(name, ax)]
            out[name] = torch.cat(
                ((W1.index_select(ax, b1) + W2.index_select(ax, b2)) / 2, W1.index_select(ax, b1c),
                 W2.index_select(ax, b2c)), ax)
        else:
            assert axes == {0, 1}
            bi1, bi2, bi1c, bi2c = blocks[Axis(name, 1)]
            bo1, bo2, bo1c, bo2c = blocks[Axis(name, 0)]
            ni, mi, no, mo = len(bi1), len(bi1c), len(bo1), len(bo1c)
            W3 = torch.zeros(no + 2 * mo, ni + 2 * mi, *W1.shape[2:])
            W3[:no, :ni] = (W1[bo1][:, bi1] + W2[bo2][:, bi2]) / 2
            W3[no:no + mo, ni:ni + mi] = W1[bo1c][:, bi1c]
            W3[no + mo:, ni + mi:] = W2[bo2c][:, bi2c]
            W3[no:no + mo, :ni] = W1[bo1c][:, bi1]
            W3[no + mo:, :ni] = W2[bo2c][:, bi2]
            W3[:no, ni:ni + mi] = W1[bo1][:, bi1c] / 2
            W3[:no, ni + mi:] = W2[bo2][:, bi2c] / 2
            out[name] = W3
    return out


def partial_merge(spec, model1, model2, perm, costs, ratios):
    """Reference: pleas/methods/partial_matching.py:188-202 -> new CPU eval-mode module."""
    blocks = get_blocks(spec, perm, costs, ratios)
    new = merged_state(spec, model1.state_dict(), model2.state_dict(), blocks)
    model3 = deepcopy(model1).eval().cpu()
    for name, w in new.items():
        set_attr(model3, name.split("."), nn.Parameter(w, requires_grad=False))
    return model3


# ============================================================================ PLeaS training
def gradient_masks(perm_blocks, layers: Dict[str, nn.Module]) -> List[torch.Tensor]:
    """Reference: pleas/methods/pleas_merging.py:11-60, including its transposed indexing
    (input slices on axis 0, output slices on axis 1, :57-58) and the arange(3)/arange(1000)
    fall-backs for axes that are not in the spec."""
    masks = []
    for name, layer in layers.items():
        bi = perm_blocks.get(Axis(name + ".weight", 1))
        bo = perm_blocks.get(Axis(name + ".weight", 0))
        ni, mi = (len(bi[0]), len(bi[2])) if bi is not None else (3, 0)
        no, mo = (len(bo[0]), len(bo[2])) if bo is not None else (1000, 0)
        for p in layer.parameters():
            m = torch.ones_like(p)
            if p.dim() >= 2:
                m[ni:ni + mi, no + mo:no + 2 * mo] = 0.0
                m[ni + mi:ni + 2 * mi, no:no + mo] = 0.0
            masks.append(m)
    return masks


def layer_targets(l1, l2, perm_blocks, name, ip1, ip2, num_classes=1000, separate_classifier=False,
                  model_type="rn50", merging="perm_gradmask"):
    """Reference: pleas/methods/pleas_merging.py:63-149: merged-layer input and regression target built from the two
    source layers.  ``merging='perm_gradmask'`` (the drivers' mode, :146-147) merges along the channel axis; the other
    modes STACK two half-batches along the sample axis (:125-144):
      reg_mean         [ip1 ; ip2]                                    ->  [o1 ; o2]           (no permutation)
      perm_separatels  [i11, i1c, 0 ; i22, 0, i2c]                    ->  [o11, o1c, 0 ; o22, 0, o2c]
      perm_mixedls     [(i11+i22)/2, i1c, 0 ; (i11+i22)/2, 0, i2c]    ->  [o11, o1c, 0 ; o22, 0, o2c]"""
    bo = perm_blocks.get(Axis(name + ".weight", 0))
    if bo is None:
        width = {"rn50": 2048, "rn101": 2048, "rn20": 1024, "rn18": 512}[model_type] if separate_classifier else num_classes
        bo = (torch.arange(width), torch.arange(width), torch.tensor([], dtype=torch.long), torch.tensor([], dtype=torch.long))
    bi = perm_blocks.get(Axis(name + ".weight", 1))
    if bi is None:
        c = ip1.shape[1]
        bi = (torch.arange(c), torch.arange(c), torch.tensor([], dtype=torch.long), torch.tensor([], dtype=torch.long))
    o1, o2 = l1(ip1), l2(ip2)
    sel = lambda t, idx: t.index_select(1, idx.long())
    if merging == "reg_mean":
        return torch.cat([ip1, ip2], 0), torch.cat([o1, o2], 0)
    i11, i22, i1c, i2c = sel(ip1, bi[0]), sel(ip2, bi[1]), sel(ip1, bi[2]), sel(ip2, bi[3])
    o11, o22, o1c, o2c = sel(o1, bo[0]), sel(o2, bo[1]), sel(o1, bo[2]), sel(o2, bo[3])
    z = torch.zeros_like
    if "perm_separatels" in merging or "perm_mixedls" in merging:
        m1_, m2_ = ((i11 + i22) / 2, (i11 + i22) / 2) if "perm_mixedls" in merging else (i11, i22)
        ip = torch.cat([torch.cat([m1_, i1c, z(i2c)], 1), torch.cat([m2_, z(i1c), i2c], 1)], 0)
        op = torch.cat([torch.cat([o11, o1c, z(o2c)], 1), torch.cat([o22, z(o1c), o2c], 1)], 0)
        return ip, op
    ip = torch.cat([(i11 + i22) / 2, i1c, i2c], 1)
    op = torch.cat([(o11 + o22) / 2, o1c, o2c], 1)
    return ip, op


def _hook_inputs(model, store):
    handles = []
    for name, mod in model.named_modules():
        if isinstance(mod, (nn.Conv2d, nn.Linear, nn.LayerNorm)):
            handles.append(mod.register_forward_hook(
                lambda m, inp, out, name=name: store.__setitem__(name, inp[0] if isinstance(inp, tuple) else inp)))
    return handles


def train(batches, model1, model2, model3, spec, perm, costs, ratios, max_steps: int, num_classes=1000, lr=5e-4,
          separate_classifier=False, model_type="rn50", merging="perm_gradmask", on_update=None, dtype=torch.float32):
    """Reference: pleas/methods/pleas_merging.py:305-405 (step: :234-302); ``merging`` as in :func:`layer_targets`.
    Adam + cosine schedule on copies of model3's Conv/Linear layers; ``max_steps + 1`` updates.
    ``dtype=torch.float64`` (with fp64 models and batches): the fp64 ANCHOR of :func:`fp64_anchor`, not the reference's arithmetic.
    ``on_update(idx, layers, per_layer_losses)`` (test aid, not in the reference): called after update ``idx`` with the
    layers being trained and this update's loss per layer, so that a trajectory can be recorded."""
    perm_blocks = spread_blocks(spec, get_blocks(spec, perm, costs, ratios))
    acts1, acts2 = {}, {}
    handles = _hook_inputs(model1, acts1) + _hook_inputs(model2, acts2)
    model1.eval()
    model2.eval()
    layers = {n: deepcopy(m).to(dtype) for n, m in model3.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))}
    params = []
    for layer in layers.values():
        for p in layer.parameters():
            p.requires_grad_(True)
            params.append(p)
    opt = torch.optim.Adam(params, lr=lr)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, max_steps)
    masks = gradient_masks(perm_blocks, layers)
    losses = []
    for idx, (x, _) in enumerate(batches):
        if idx > max_steps:
            break
        with torch.no_grad():
            model1(x)
            model2(x)
        opt.zero_grad()
        total = 0.0
        per_layer = []
        for name, layer in layers.items():
            with torch.no_grad():
                ip, op = layer_targets(get_attr(model1, name.split(".")), get_attr(model2, name.split(".")),
                                       perm_blocks, name, acts1[name], acts2[name], num_classes,
                                       separate_classifier, model_type, merging)
            loss = ((layer(ip) - op) ** 2).mean()
            per_layer.append(float(loss.detach()))
            total = total + loss
        total.backward()
        for p, m in zip(params, masks):
            p.grad *= m
        opt.step()
        sched.step()
        losses.append(float(total.detach()))
        if on_update is not None:
            on_update(idx, layers, per_layer)
        acts1.clear()
        acts2.clear()
    sd = model3.state_dict()
    for name, layer in layers.items():
        for k, v in layer.state_dict().items():
            sd["%s.%s" % (name, k)] = v.detach()
    model3.load_state_dict(sd)
    for h in handles:
        h.remove()
    return model3, losses


def fp64_anchor(spec, model1, model2, batches, n_match: int, n_updates: int, perm, costs, ratios, max_steps: int,
                num_classes=1000):
    """The SAME path in fp64 on the same batches: what the reference's fp32 arithmetic (pleas_merging.py:281-291,
    activation_matching.py:119-134) and any second fp32 implementation of it both approximate.  Distances TO this anchor are
    statements about accuracy; distances between two fp32 runs are statements about spread.
    Matching costs: fp64 models and batches through :func:`matching_costs`.  Training: the fp32-merged model (the merge is
    exact and shared by both sides) cast to fp64, fp64 sources, ``n_updates`` Adam updates from ``perm`` / ``costs``.
    Returns (costs64 or None, trained fp64 state dict or None)."""
    d1, d2 = deepcopy(model1).double().eval(), deepcopy(model2).double().eval()
    data = [(x.double(), y) for x, y in batches[:max(n_match, n_updates)]]
    costs64 = matching_costs(spec, d1, d2, data[:n_match], n_match, accumulate=True) if n_match > 0 else None
    trained = None
    if n_updates > 0:
        o3 = partial_merge(spec, model1, model2, perm, costs, ratios).double()
        o3, _ = train(data[:n_updates], d1, d2, o3, spec, perm, costs, ratios, max_steps, num_classes=num_classes,
                      dtype=torch.float64)
        trained = {k: v.clone() for k, v in o3.state_dict().items()}
    return costs64, trained


# ============================================================================ closed form (north star)
def unfold_rows(ip: torch.Tensor, layer: nn.Module) -> torch.Tensor:
    """im2col of a layer input: rows = samples x output pixels, columns = Cin*kh*kw (+1 if bias)."""
    if isinstance(layer, nn.Conv2d):
        U = F.unfold(ip, layer.kernel_size, layer.dilation, layer.padding, layer.stride)  # B, K, L
        U = U.transpose(1, 2).reshape(-1, U.shape[1])
    else:
        U = ip.reshape(-1, ip.shape[-1])
    if layer.bias is not None:
        U = torch.cat([U, torch.ones(U.shape[0], 1, dtype=U.dtype)], 1)
    return U


def normal_equations(ips: Sequence[torch.Tensor], ops: Sequence[torch.Tensor], layer: nn.Module):
    """A = sum U^T U, B = sum U^T Y in fp64 (objective of pleas_merging.py:281-284, SURVEY.md 8(a) tail)."""
    A = Bm = None
    for ip, op in zip(ips, ops):
        U = unfold_rows(ip.double(), layer)
        Y = op.double().reshape(op.shape[0], op.shape[1], -1).transpose(1, 2).reshape(-1, op.shape[1]) \
            if op.dim() == 4 else op.double().reshape(-1, op.shape[-1])
        A = U.T @ U if A is None else A + U.T @ U
        Bm = U.T @ Y if Bm is None else Bm + U.T @ Y
    return A, Bm


def solve_normal_equations(A: torch.Tensor, Bm: torch.Tensor, ridge: float = 0.0) -> torch.Tensor:
    """W^T = (A + ridge*mean(diag A)*I)^-1 B via fp64 lstsq -> (K, Cout)."""
    K = A.shape[0]
    A = A + ridge * A.diagonal().mean() * torch.eye(K, dtype=A.dtype)
    return torch.linalg.lstsq(A, Bm).solution
