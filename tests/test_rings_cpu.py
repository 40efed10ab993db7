"""Book-keeping of the two HBM replay rings (replay/rings.py) without a GPU: a host class that records the launches the
mixin would make. FIFO eviction of whole blocks across both rings, ranges that start over at 0, blocks evicted while their
data is still in flight, and the coalescing the ingest pipeline relies on (engine/ingest.py: one retire launch per run of
neighbouring evicted blocks, one commit launch per run of neighbouring written blocks)."""
import random

import pytest

from distributed_rl_amd.replay.rings import RingPair


class _Host(RingPair):
    def __init__(self, capacity, plane_capacity):
        self.capacity, self.plane_capacity = capacity, plane_capacity
        self.retired, self.committed = [], []
        self._init_rings()

    def _retire(self, slot, K):
        self.retired.append((slot, K))

    def _commit_range(self, slot, n):
        self.committed.append((slot, n))


def _live(h):
    return [(b.slot, b.K, b.plane, b.M) for b in h._blocks if b.state == 1]


def test_deferred_retires_of_neighbouring_blocks_share_a_launch():
    h = _Host(100, 1000)
    blocks = [h.reserve_block(10, 12) for _ in range(10)]            # fills the item ring exactly
    assert h.commit_blocks(blocks) == 1 and h.committed == [(0, 100)] and h.count == 100
    ev = []
    b1 = h.reserve_block(10, 12, ev)                                  # slot 0 again: evicts the oldest block
    b2 = h.reserve_block(25, 30, ev)                                  # the next three
    assert ev == [(0, 10), (10, 10), (20, 10), (30, 10)] and h.retired == [] and h.count == 60
    assert h.retire_ranges(ev) == 1 and h.retired == [(0, 40)]
    assert (b1.slot, b2.slot) == (0, 10) and h.evicted_items == 40
    # immediate form (no list): one launch per evicted block, as before
    h.reserve_block(10, 12)
    assert h.retired[1:] == [(40, 10)]


def test_commit_runs_break_at_gaps_and_skip_blocks_evicted_in_flight():
    h = _Host(64, 640)
    a = h.reserve_block(16, 20)
    b = h.reserve_block(16, 20)
    c = h.reserve_block(16, 20)
    d = h.reserve_block(16, 20)
    e = h.reserve_block(16, 20, [])                                   # wraps: takes a's range while a is still in flight
    assert a.state == 2 and e.slot == 0 and h.retired == []          # nothing of a was in the tree: nothing to retire
    assert h.commit_blocks([a, b, c]) == 1 and h.committed == [(16, 32)] and h.count == 32
    assert h.commit_blocks([d, e]) == 2 and h.committed[1:] == [(48, 16), (0, 16)] and h.count == 64
    assert [x.state for x in (a, b, c, d, e)] == [2, 1, 1, 1, 1]


def test_a_range_that_does_not_fit_starts_over_and_the_skipped_tail_counts_as_overwritten():
    h = _Host(50, 70)
    for _ in range(3):
        h.commit_blocks([h.reserve_block(12, 20)])                    # slots 0..35, planes 0..59
    ev = []
    b = h.reserve_block(12, 20, ev)                                   # slots 36..47 fit, planes 60..79 do not: planes restart at 0
    assert (b.slot, b.plane) == (36, 0) and ev == [(0, 12)]           # the block that owned planes 0..19 goes
    b2 = h.reserve_block(12, 20, ev)                                  # slots 48..59 do not fit: slots restart at 0
    assert (b2.slot, b2.plane) == (0, 20) and ev == [(0, 12), (12, 12)]
    assert h.retire_ranges(ev) == 1 and h.retired == [(0, 24)]


def test_random_traffic_keeps_the_rings_in_ingestion_order():
    rng = random.Random(7)
    h = _Host(500, 700)
    inflight = []
    for step in range(3000):
        K = rng.randint(1, 60)
        M = K + rng.randint(3, 10)
        ev = []
        inflight.append(h.reserve_block(K, M, ev))
        h.retire_ranges(ev)
        if len(inflight) > rng.randint(0, 6):                         # commits lag the reservations by a few blocks
            n = rng.randint(1, len(inflight))
            h.commit_blocks(inflight[:n])
            inflight = inflight[n:]
        live = _live(h)
        assert sum(k for _, k, _, _ in live) == h.count <= h.capacity
        # no two live or in-flight blocks share an item slot or a plane
        owned = [(b.slot, b.slot + b.K, b.plane, b.plane + b.M) for b in h._blocks if b.state != 2]
        for i, (s0, s1, p0, p1) in enumerate(owned):
            assert s1 <= h.capacity and p1 <= h.plane_capacity
            for t0, t1, q0, q1 in owned[i + 1:]:
                assert s1 <= t0 or t1 <= s0
                assert p1 <= q0 or q1 <= p0
    assert h.evicted_items > 10 * h.capacity                          # the rings lapped many times
    with pytest.raises(ValueError):
        h.reserve_block(501, 10)
