"""Book-keeping shared by the two HBM replays (replay/device_per.py, replay/device_seq.py): an ITEM ring (transition or
sequence slots, with the sum-tree over it) and a PLANE ring (single 84x84 uint8 frames, each stored once) that advance
together in ingestion order.

A block - what one actor publication carries - owns one contiguous range of each ring. A range that does not fit before
the end of its ring starts over at 0 and the skipped tail counts as overwritten, so both rings stay in ingestion order and
the blocks in the way of a new one are always the OLDEST (front of the deque). Before a range is handed out, every older
block that still owns part of it is RETIRED as a whole: its items leave the tree (priority 0, generation bumped), so
nothing samples an item whose planes are gone. That is the reference's FIFO eviction (``remove_to_fit``,
libs/replay_memory.py:70-75), applied immediately instead of every 100 learner steps.
"""
from __future__ import annotations

from collections import deque
from typing import Deque


class Block:
    """state: 0 = reserved (its data may still be on the way: engine/ingest.py keeps several blocks in flight), 1 =
    committed (in the tree), 2 = evicted before its commit (a replay smaller than what is in flight): the commit is then
    dropped."""
    __slots__ = ("slot", "K", "plane", "M", "state")

    def __init__(self, slot: int, K: int, plane: int, M: int):
        self.slot, self.K, self.plane, self.M, self.state = slot, K, plane, M, 0


def _overlap(a0: int, a1: int, b0: int, b1: int) -> bool:
    return a0 < b1 and b0 < a1


class RingPair:
    """Mixin. The host class provides ``capacity`` (items), ``plane_capacity``, ``_retire(slot, K)`` (kernel launch that
    takes items out of the tree), ``_commit_range(slot, n)`` (launch that puts items whose priorities wait in its
    ``prio_pending`` into the tree) and keeps ``count`` (live items)."""

    def _init_rings(self) -> None:
        self.cursor = 0              # next item slot
        self.pcursor = 0             # next plane
        self.count = 0               # live items (host view)
        self.evicted_items = 0       # items retired to make room (FIFO)
        self._blocks: Deque[Block] = deque()

    def reserve_block(self, K: int, M: int, deferred=None) -> Block:
        """``deferred`` (a list): the evicted item ranges are appended to it instead of being retired one launch each; the
        caller hands the list to ``retire_ranges`` before anything can sample again (engine/ingest.py reserves all blocks
        of a service call, then retires once)."""
        if K > self.capacity or M > self.plane_capacity:
            raise ValueError("block larger than the replay (%d items / %d planes; capacity %d / %d)" %
                             (K, M, self.capacity, self.plane_capacity))
        slot = self.cursor if self.cursor + K <= self.capacity else 0
        plane = self.pcursor if self.pcursor + M <= self.plane_capacity else 0
        s_rng = [(slot, slot + K)] + ([(self.cursor, self.capacity)] if slot != self.cursor else [])
        p_rng = [(plane, plane + M)] + ([(self.pcursor, self.plane_capacity)] if plane != self.pcursor else [])

        def in_the_way(b: Block) -> bool:
            return any(_overlap(b.slot, b.slot + b.K, lo, hi) for lo, hi in s_rng) or \
                any(_overlap(b.plane, b.plane + b.M, lo, hi) for lo, hi in p_rng)

        retired = 0
        while self._blocks and in_the_way(self._blocks[0]):
            b = self._blocks.popleft()
            if b.state == 1:
                self.count -= b.K
                if deferred is None:
                    self._retire(b.slot, b.K)                     # FIFO eviction, whole blocks (the launch also lowers the
                else:                                             # device copy of the item count: IS weights read it)
                    deferred.append((b.slot, b.K))
                retired += b.K
            b.state = 2
        self.evicted_items += retired
        if slot != self.cursor or plane != self.pcursor:          # once per lap: the order argument, checked
            assert not any(in_the_way(b) for b in self._blocks), "replay rings out of ingestion order"
        self.cursor, self.pcursor = slot + K, plane + M
        blk = Block(slot, K, plane, M)
        self._blocks.append(blk)                                  # owner of its ranges from now on
        return blk

    def retire_ranges(self, ranges) -> int:
        """Retire the item ranges collected by ``reserve_block(deferred=...)``: neighbours in the ring (evictions are FIFO,
        so they nearly always are) share one launch. Returns the number of launches."""
        n_launch, run_s, run_n = 0, 0, 0
        for slot, K in list(ranges) + [(-1, 0)]:
            if run_n and slot == run_s + run_n:
                run_n += K
                continue
            if run_n:
                self._retire(run_s, run_n)
                n_launch += 1
            run_s, run_n = slot, K
        return n_launch

    def commit_blocks(self, blks) -> int:
        """Commit blocks written by ``write_staged`` (their priorities wait in ``prio_pending`` at their own slots): blocks
        that follow each other in the item ring - the usual case, they were reserved in that order - share ONE launch.
        Blocks evicted while in flight (state 2) are dropped. Returns the number of launches."""
        n_launch, run_s, run_n = 0, 0, 0
        for blk in list(blks) + [None]:
            if blk is not None and blk.state == 2:
                continue
            if blk is not None and run_n and blk.slot == run_s + run_n:
                run_n += blk.K
            else:
                if run_n:
                    self.count += run_n
                    self._commit_range(run_s, run_n)
                    n_launch += 1
                run_s, run_n = (blk.slot, blk.K) if blk is not None else (0, 0)
            if blk is not None:
                blk.state = 1
        return n_launch
