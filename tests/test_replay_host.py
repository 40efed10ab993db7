"""Sum-tree / PrioritizedMemory semantics (reference: libs/sumtree.py, libs/replay_memory.py;
untested upstream — SURVEY §4)."""
import numpy as np
import pytest

from distributed_rl_amd.replay.memory import PrioritizedMemory
from distributed_rl_amd.replay.sumtree import SumTree


def test_sumtree_deque_semantics():
    t = SumTree(capacity=2)
    vals = [float(i + 1) for i in range(37)]
    for v in vals:
        t.append(v)                       # grows by doubling (sumtree.py:80-95)
    assert len(t) == 37 and abs(t.total - sum(vals)) < 1e-9
    assert t[0] == 1.0 and t[36] == 37.0
    assert t.popleft() == 1.0             # sumtree.py:101-118
    assert t[0] == 2.0 and len(t) == 36 and abs(t.total - (sum(vals) - 1)) < 1e-9
    t[5] = 100.0
    assert t[5] == 100.0
    assert abs(t.total - (sum(vals) - 1 - 7 + 100)) < 1e-9
    t.popleft(10)
    t.extend(np.arange(50.0))
    assert abs(t.total - t.leaves().sum()) < 1e-9
    with pytest.raises(IndexError):
        t[len(t)]


def test_sumtree_no_drift():
    rng = np.random.default_rng(0)
    t = SumTree(1024, rng=rng)
    t.extend(rng.random(1000))
    for _ in range(200):
        ix = rng.integers(0, len(t), 64)
        t[ix] = rng.random(64) * 10
    assert t.total == pytest.approx(float(t.leaves().sum()), rel=1e-12)


def test_sumtree_sampling_distribution():
    rng = np.random.default_rng(1)
    t = SumTree(64, rng=rng)
    p = np.array([1, 2, 3, 4, 0, 10, 0.5, 0], dtype=np.float64)
    t.extend(p)
    ix, pr = t.prioritized_sample(40000)
    cnt = np.bincount(ix, minlength=len(p))
    assert cnt[4] == 0 and cnt[7] == 0                 # never a zero-priority leaf
    exp = p / p.sum() * 40000
    chi2 = ((cnt - exp) ** 2 / np.maximum(exp, 1e-9))[p > 0].sum()
    assert chi2 < 30                                    # 5 dof, generous
    assert np.allclose(pr, p[ix])
    ix, _ = t.prioritized_sample(8, stratified=True)
    assert len(ix) == 8


def test_sumtree_duplicate_update_last_wins():
    t = SumTree(16)
    t.extend(np.ones(8))
    m = PrioritizedMemory(100)
    m.push(list(range(8)), [1.0] * 8)
    m.update_priorities([3, 3, 5], [7.0, 9.0, 2.0])
    assert m.priorities[3] == 9.0 and m.priorities[5] == 2.0


def test_prioritized_memory_soft_capacity_and_ids():
    m = PrioritizedMemory(10, rng=np.random.default_rng(0))
    m.push(list(range(25)), [1.0] * 25)
    assert len(m) == 25                                   # soft capacity (replay_memory.py:58-60)
    items, prios, idxs = m.sample(5)
    assert all(items[k] == idxs[k] for k in range(5))
    ids = m.ids_of(idxs)
    assert m.remove_to_fit() == 15 and len(m) == 10       # FIFO eviction (replay_memory.py:70-75)
    items, _, idxs2 = m.sample(5)
    assert all(items[k] == idxs2[k] + 15 for k in range(5))
    # stale write-backs (items evicted since sampling) are dropped, live ones land on the right item
    applied = m.update_priorities(idxs, [5.0] * 5, ids=ids)
    live = [i for i in ids if i >= 15]
    assert applied == len(live)
    for i in live:
        assert m.priorities[i - 15] == 5.0
    assert m.total_prios() == pytest.approx(m.priorities.leaves().sum())


def test_prioritized_memory_compress():
    m = PrioritizedMemory(10, use_compress=True)
    m.push([("a", i) for i in range(4)], [1, 2, 3, 4])
    items, _, idxs = m.sample(3)
    assert all(it == ("a", ix) for it, ix in zip(items, idxs))
