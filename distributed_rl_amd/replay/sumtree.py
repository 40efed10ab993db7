"""Host sum-tree with deque semantics (reference: libs/sumtree.py:3-138).

The reference is a pointer-based, growable binary tree whose index window slides on ``popleft``.
This one is an implicit array tree over a power-of-two ring: logical index ``i`` (0 = oldest)
lives in leaf ``(head + i) mod cap``; ``append`` doubles the ring when full, ``popleft`` zeroes the
oldest leaf and advances ``head``. Parents are always recomputed from their two children (never
incrementally adjusted), so there is no float drift (sumtree.py:20-26). All batch operations are
vectorised numpy; this class is also the CPU oracle for the HIP sum-tree kernels.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import numpy as np


class SumTree:
    def __init__(self, capacity: int = 1024, rng: Optional[np.random.Generator] = None, dtype=np.float64):
        cap = 1
        while cap < max(2, capacity):
            cap *= 2
        self._cap = cap
        self._tree = np.zeros(2 * cap, dtype=dtype)    # node 1 = root, leaves at [cap, 2cap)
        self._head = 0
        self.length = 0
        self._rng = rng if rng is not None else np.random.default_rng()

    # ---- structure --------------------------------------------------------------------------------
    def __len__(self) -> int:
        return self.length

    @property
    def capacity(self) -> int:
        return self._cap

    @property
    def total(self) -> float:
        return float(self._tree[1])

    def _phys(self, ix):
        return (self._head + np.asarray(ix, dtype=np.int64)) & (self._cap - 1)

    def _grow(self) -> None:
        old = self._tree[self._cap:2 * self._cap]
        vals = np.concatenate([old[self._head:], old[:self._head]])[:self.length]
        self._cap *= 2
        self._tree = np.zeros(2 * self._cap, dtype=self._tree.dtype)
        self._head = 0
        self._tree[self._cap:self._cap + self.length] = vals
        self._rebuild()

    def _rebuild(self) -> None:
        n = self._cap
        while n > 1:
            lv = self._tree[n:2 * n]
            self._tree[n // 2:n] = lv[0::2] + lv[1::2]
            n //= 2

    def _set_leaves(self, phys: np.ndarray, vals: np.ndarray) -> None:
        """Scatter leaves (last write wins for duplicates) and recompute the touched ancestors level by level."""
        node = phys + self._cap
        self._tree[node] = vals
        node = np.unique(node >> 1)
        while node.size and node[0] >= 1:
            self._tree[node] = self._tree[2 * node] + self._tree[2 * node + 1]
            if node[0] == 1:
                break
            node = np.unique(node >> 1)

    # ---- deque API (sumtree.py:69-121) ------------------------------------------------------------
    def __setitem__(self, ix, val) -> None:
        ix = np.atleast_1d(np.asarray(ix, dtype=np.int64))
        val = np.broadcast_to(np.asarray(val, dtype=self._tree.dtype), ix.shape)
        if ix.size and (ix.min() < 0 or ix.max() >= self.length):
            raise IndexError("sum-tree index out of range")
        self._set_leaves(self._phys(ix), val)

    def __getitem__(self, ix):
        if np.isscalar(ix):
            if not 0 <= ix < self.length:
                raise IndexError("sum-tree index out of range")
            return float(self._tree[self._cap + int(self._phys(ix))])
        return self._tree[self._cap + self._phys(ix)]

    def append(self, value: float) -> None:
        self.extend([value])

    def extend(self, values: Iterable[float]) -> None:
        vals = np.asarray(list(values) if not isinstance(values, np.ndarray) else values, dtype=self._tree.dtype)
        if vals.size == 0:
            return
        while self.length + vals.size > self._cap:
            self._grow()
        phys = self._phys(np.arange(self.length, self.length + vals.size))
        self.length += vals.size
        self._set_leaves(phys, vals)

    def popleft(self, n: int = 1):
        assert 0 < n <= self.length
        phys = self._phys(np.arange(n))
        ret = self._tree[self._cap + phys].copy()
        self._set_leaves(phys, np.zeros(n, dtype=self._tree.dtype))
        self._head = (self._head + n) & (self._cap - 1)
        self.length -= n
        return float(ret[0]) if n == 1 else ret

    # ---- sampling (sumtree.py:127-138) ------------------------------------------------------------
    def find(self, pos: np.ndarray) -> np.ndarray:
        """Descend for each prefix mass in ``pos`` → *logical* index. Never lands on a zero leaf:
        if float edge effects push a query past the occupied mass, go left (the reference crashes
        there, sumtree.py:57-61)."""
        pos = np.asarray(pos, dtype=self._tree.dtype).copy()
        node = np.ones(pos.shape, dtype=np.int64)
        while node[0] < self._cap if node.size else False:
            left = 2 * node
            lv = self._tree[left]
            rv = self._tree[left + 1]
            go_right = (pos >= lv) & (rv > 0)
            pos = np.where(go_right, pos - lv, pos)
            node = np.where(go_right, left + 1, left)
        return ((node - self._cap) - self._head) & (self._cap - 1)

    def prioritized_sample(self, n: int, stratified: bool = False) -> Tuple[List[int], List[float]]:
        """n independent draws ∝ priority, with replacement (sumtree.py:133-137).
        ``stratified=True`` draws one sample per equal-mass stratum instead."""
        assert n >= 0
        if n == 0 or self.length == 0:
            return [], []
        total = self.total
        if stratified:
            u = (np.arange(n) + self._rng.random(n)) * (total / n)
        else:
            u = self._rng.uniform(0.0, total, size=n)
        ix = self.find(u)
        ix = np.minimum(ix, self.length - 1)
        return ix.tolist(), self._tree[self._cap + self._phys(ix)].tolist()

    def leaves(self) -> np.ndarray:
        """Priorities in logical order (oldest first)."""
        return self._tree[self._cap + self._phys(np.arange(self.length))].copy()
