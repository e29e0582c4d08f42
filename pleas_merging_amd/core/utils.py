"""Data model of the matching/merging path: permutable axes, groups, permutations.

Mirrors the public names of the reference's ``pleas/core/utils.py`` (reference
file:line cited per symbol) so that callers of ``pleas.core`` can switch without
edits.  Only what the hot path needs is here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, Mapping, Optional, Sequence, Union

import torch
from torch import nn


@dataclass(frozen=True)
class Axis:
    """One axis of a named tensor (state-dict key or fx node name).

    Reference: pleas/core/utils.py:16-31 (frozen, hashable, ``str`` = "key:axis").
    """

    key: str
    axis: int

    def __str__(self) -> str:
        return "%s:%d" % (self.key, self.axis)

    __repr__ = __str__

    @staticmethod
    def parse(text: str) -> "Axis":
        key, _, ax = text.rpartition(":")
        return Axis(key, int(ax))


@dataclass
class PermutationGroup:
    """Axes that must be permuted together (reference: pleas/core/utils.py:34-46)."""

    size: int
    state: set = field(default_factory=set)
    node: set = field(default_factory=set)


PermutationKey = Axis
PermutationSpec = Dict[Axis, PermutationGroup]
Permutation = Dict[Axis, torch.Tensor]
StateDict = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- attr helpers
def get_attr(obj, names: Sequence[str]):
    """Nested ``getattr`` along ``names`` (reference: pleas/core/utils.py:108-122)."""
    for name in names:
        obj = getattr(obj, name)
    return obj


def set_attr(obj, names: Sequence[str], val) -> None:
    """Nested ``setattr`` along ``names`` (reference: pleas/core/utils.py:125-140)."""
    names = list(names)
    setattr(get_attr(obj, names[:-1]), names[-1], val)


# --------------------------------------------------------------------------- permutations
def make_identity_perm(spec: PermutationSpec) -> Permutation:
    """Reference: pleas/core/utils.py:143-153."""
    return {key: torch.arange(group.size) for key, group in spec.items()}


def make_random_perm(spec: PermutationSpec, generator: Optional[torch.Generator] = None) -> Permutation:
    """Reference: pleas/core/utils.py:156-166 (``generator`` is an addition for seeded tests)."""
    return {key: torch.randperm(group.size, generator=generator) for key, group in spec.items()}


def invert_perm(perm: Union[torch.Tensor, Permutation]):
    """Inverse permutation, elementwise over a dict (reference: pleas/core/utils.py:169-184)."""
    if isinstance(perm, Mapping):
        return {key: invert_perm(p) for key, p in perm.items()}
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), dtype=perm.dtype, device=perm.device)
    return inv


def perm_eq(perm1: Permutation, perm2: Permutation) -> bool:
    """Reference: pleas/core/utils.py:187-200."""
    if len(perm1) != len(perm2):
        return False
    return all(bool((perm2[key].cpu() == p.cpu()).all()) for key, p in perm1.items())


def apply_perm(
    perm: Permutation,
    spec: PermutationSpec,
    state: Union[nn.Module, StateDict],
    inplace: bool = False,
    skip_missing: bool = True,
):
    """Gather every state axis of each permuted group by its permutation.

    Reference: pleas/core/utils.py:203-246.  A module must be passed with
    ``inplace=True`` (same assertion as the reference); ``inplace`` on a dict
    rebinds entries of that dict, it never writes into the tensors.
    """
    if isinstance(state, nn.Module):
        assert inplace is True, "apply_perm on a module needs inplace=True"
        state.load_state_dict(apply_perm(perm, spec, state.state_dict(), inplace=True, skip_missing=skip_missing))
        return state

    out = state if inplace else dict(state)
    for key, p in perm.items():
        if p is None:
            continue
        group = spec[key]
        assert tuple(p.shape) == (group.size,), (key, tuple(p.shape), group.size)
        for ax in group.state:
            if ax.key not in out:
                if skip_missing:
                    continue
                raise KeyError(ax.key)
            w = out[ax.key]
            out[ax.key] = torch.index_select(w, ax.axis, p.to(w.device))
    return out


# --------------------------------------------------------------------------- union-find
class UnionFind:
    """Disjoint sets over hashable items; remembers first-insertion order.

    Plays the role of the reference's ``UnionFind`` (pleas/core/utils.py:331-380);
    ``groups()`` lists sets in the order their first member was inserted, which is
    what fixes the order of a ``PermutationSpec``.
    """

    def __init__(self, items: Iterable = ()):
        self._parent: dict = {}
        self._rank: dict = {}
        for it in items:
            self.add(it)

    def add(self, item) -> None:
        if item not in self._parent:
            self._parent[item] = item
            self._rank[item] = 0

    def find(self, item, add: bool = False):
        if add:
            self.add(item)
        root = item
        while self._parent[root] != root:
            root = self._parent[root]
        while self._parent[item] != root:  # path compression
            self._parent[item], item = root, self._parent[item]
        return root

    def union(self, a, b, add: bool = False) -> None:
        ra, rb = self.find(a, add), self.find(b, add)
        if ra == rb:
            return
        if self._rank[ra] < self._rank[rb]:
            ra, rb = rb, ra
        self._parent[rb] = ra
        if self._rank[ra] == self._rank[rb]:
            self._rank[ra] += 1

    def groups(self) -> list:
        out: dict = {}
        for item in self._parent:
            out.setdefault(self.find(item), []).append(item)
        return list(out.values())


# --------------------------------------------------------------------------- spec (de)serialisation
def spec_to_json(spec: PermutationSpec) -> list:
    """Order-preserving plain-data form of a spec (used by fixtures and save/load)."""
    return [
        {
            "key": str(key),
            "size": int(group.size),
            "state": sorted(str(a) for a in group.state),
            "node": sorted(str(a) for a in group.node),
        }
        for key, group in spec.items()
    ]


def spec_from_json(rows: list) -> PermutationSpec:
    return {
        Axis.parse(r["key"]): PermutationGroup(
            int(r["size"]), {Axis.parse(a) for a in r["state"]}, {Axis.parse(a) for a in r["node"]}
        )
        for r in rows
    }
