"""Replay service: ingest thread + locked PER facade (reference: ape_x/replay.py:7-54).

A daemon thread drains the experience channel every ``poll`` seconds and appends to the replay
under a lock; ``sample / update_priorities / remove_to_fit`` take the same lock. Unlike the
reference, ``sample`` also returns a consistent ``(len, total_prio)`` snapshot taken under the
lock (the reference reads them unlocked afterwards, replay.py:49-54 — SURVEY §5.2) and item ids so
that stale priority write-backs are dropped.

The same class fronts either the host :class:`~distributed_rl_amd.replay.memory.PrioritizedMemory`
(CPU plumbing path) or a device replay (``replay.device_per``), which exposes ``ingest_block``.
"""
from __future__ import annotations

import threading
import time
from typing import List, Optional

import numpy as np

from ..replay.memory import PrioritizedMemory
from ..transport import ExperienceChannel, codec


def block_to_items(block) -> List:
    """Split a decoded block into per-item tuples for the host PER (copies out of the ring slot)."""
    if block["kind"] == codec.KIND_APEX_PLANES:                  # plane pool -> the stacks the host PER stores
        K, pl = block["action"].shape[0], block["planes"]
        st = lambda w: np.ascontiguousarray(np.transpose(pl[w:w + 4], (1, 2, 0)))   # noqa: E731
        return [(st(int(block["win_s"][i])), int(block["action"][i]), float(block["reward"][i]),
                 st(int(block["win_s2"][i])), float(block["done"][i])) for i in range(K)]
    if block["kind"] == codec.KIND_APEX:
        K = block["action"].shape[0]
        return [(block["state"][i].copy(), int(block["action"][i]), float(block["reward"][i]),
                 block["next_state"][i].copy(), float(block["done"][i])) for i in range(K)]
    K = block["action"].shape[0]
    if block["kind"] == codec.KIND_R2D2_PLANES:                  # plane pool -> the per-step stacks the host PER stores
        from ..algo.vec_assembly import stacks_from_planes
        return [dict(frames=stacks_from_planes(block["planes"], block["win"][i]), action=block["action"][i].copy(),
                     reward=block["reward"][i].copy(), done=block["done"][i].copy(), length=int(block["length"][i]),
                     h0=block["h0"][i].copy(), c0=block["c0"][i].copy()) for i in range(K)]
    return [dict(frames=block["frames"][i].copy(), action=block["action"][i].copy(), reward=block["reward"][i].copy(),
                 done=block["done"][i].copy(), length=int(block["length"][i]), h0=block["h0"][i].copy(),
                 c0=block["c0"][i].copy()) for i in range(K)]


class Replay(threading.Thread):
    def __init__(self, size: int, channel: ExperienceChannel, memory=None, poll: float = 0.01,
                 max_blocks_per_poll: int = 64):
        super().__init__(daemon=True, name="replay-ingest")
        self._memory = memory if memory is not None else PrioritizedMemory(size)
        self._channel = channel
        self._lock = threading.Lock()
        self._poll = poll
        self._max_blocks = max_blocks_per_poll
        self._stop_evt = threading.Event()
        self.ingested_items = 0
        self.ingested_blocks = 0
        self.env_steps_seen = 0

    # ---- ingest thread ----------------------------------------------------------------------------
    def ingest_once(self) -> int:
        blocks = self._channel.drain(self._max_blocks)
        n = 0
        for b in blocks:
            with self._lock:
                if hasattr(self._memory, "ingest_block"):
                    k = self._memory.ingest_block(b)
                else:
                    items = block_to_items(b)
                    self._memory.push(items, b["prio"].tolist())
                    k = len(items)
            n += k
            self.ingested_blocks += 1
        self.ingested_items += n
        return n

    def run(self):
        while not self._stop_evt.is_set():
            try:
                got = self.ingest_once()
            except Exception:  # keep the service alive; the learner notices via len() stalling
                import traceback
                traceback.print_exc()
                got = 0
            if got == 0:
                time.sleep(self._poll)

    def stop(self):
        self._stop_evt.set()

    # ---- locked facade ----------------------------------------------------------------------------
    def sample(self, batch_size: int):
        """→ (items, prios, idxs, ids, n_items, total_prio) — one consistent snapshot."""
        with self._lock:
            items, prios, idxs = self._memory.sample(batch_size)
            ids = self._memory.ids_of(idxs) if hasattr(self._memory, "ids_of") else idxs
            return items, prios, idxs, ids, len(self._memory), self._memory.total_prios()

    def update_priorities(self, indices, priorities, ids=None):
        with self._lock:
            if ids is not None:
                return self._memory.update_priorities(indices, priorities, ids=ids)
            return self._memory.update_priorities(indices, priorities)

    def remove_to_fit(self):
        with self._lock:
            return self._memory.remove_to_fit()

    def __len__(self):
        return len(self._memory)

    @property
    def total_prios(self):
        return self._memory.total_prios()

    @property
    def memory(self):
        return self._memory
