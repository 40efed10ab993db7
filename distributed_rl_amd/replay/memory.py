"""Host prioritized replay of the PyTorch learner path (reference: libs/replay_memory.py:51-81).

:class:`PrioritizedMemory` — items + sum-tree with the reference's *soft capacity*: it may grow past ``capacity`` until
:meth:`remove_to_fit` pops the oldest overflow (replay_memory.py:70-75). The reference's uniform ``ReplayMemory`` (the
actors' staging buffer) and its lz4 ``CompressedDeque`` have no counterpart here: actors stage in preallocated arrays
(runtime/actor.py) and frames travel as raw uint8 planes.

Indices handed out by ``PrioritizedMemory.sample`` are *logical* (0 = oldest) like the reference's,
and in addition every item carries a monotonically increasing id; :meth:`update_priorities`
accepts ``ids=`` so a write-back that races with an eviction is dropped instead of landing on the
wrong item (the generation-tag invariant of SURVEY §5.2).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import utils
from .sumtree import SumTree


class PrioritizedMemory:
    def __init__(self, capacity: int, use_compress: bool = False, rng: Optional[np.random.Generator] = None):
        self.capacity = capacity
        self.use_compress = use_compress
        self.transitions: List = []          # ring-free list + head offset: O(1) index, amortised O(1) pop
        self._head = 0
        self._first_id = 0                   # id of logical index 0
        self.priorities = SumTree(min(max(capacity, 2), 1 << 16), rng=rng)

    def _enc(self, item):
        return utils.dumps(item) if self.use_compress else item

    def _dec(self, item):
        return utils.loads(item) if self.use_compress else item

    def push(self, transitions: Sequence, priorities: Sequence[float]) -> None:
        assert len(transitions) == len(priorities)
        self.transitions.extend(self._enc(t) for t in transitions)
        self.priorities.extend(np.asarray(priorities, dtype=np.float64))

    def sample(self, batch_size: int) -> Tuple[list, List[float], List[int]]:
        idxs, prios = self.priorities.prioritized_sample(batch_size)
        return [self._dec(self.transitions[self._head + i]) for i in idxs], prios, idxs

    def ids_of(self, indices: Sequence[int]) -> List[int]:
        return [self._first_id + i for i in indices]

    def update_priorities(self, indices: Sequence[int], priorities: Sequence[float],
                          ids: Optional[Sequence[int]] = None) -> int:
        """Set priorities; with ``ids`` (from :meth:`ids_of` at sample time) stale entries are dropped.
        Duplicate indices: last write wins (the reference's sequential loop, replay_memory.py:66-68).
        Returns the number of updates applied."""
        idx = np.asarray(indices, dtype=np.int64)
        pr = np.asarray(priorities, dtype=np.float64)
        if ids is not None:
            idx = np.asarray(ids, dtype=np.int64) - self._first_id
        ok = (idx >= 0) & (idx < len(self))
        idx, pr = idx[ok], pr[ok]
        if idx.size:
            # keep the last occurrence of each duplicate
            _, last = np.unique(idx[::-1], return_index=True)
            sel = idx.size - 1 - last
            self.priorities[idx[sel]] = pr[sel]
        return int(idx.size)

    def remove_to_fit(self) -> int:
        over = len(self) - self.capacity
        if over <= 0:
            return 0
        self.priorities.popleft(over)
        self._head += over
        self._first_id += over
        if self._head > 4096 and self._head > len(self.transitions) // 2:
            del self.transitions[:self._head]
            self._head = 0
        return over

    def __len__(self) -> int:
        return len(self.transitions) - self._head

    def total_prios(self) -> float:
        return self.priorities.total
