"""n-step / sequence assembly (ape_x/actor.py:60-90, r2d2/actor.py:26-75) and the synthetic env contract
(wrapped_env.py:30-47)."""
import numpy as np
import pytest

from distributed_rl_amd import envs
from distributed_rl_amd.algo.nstep import NStepAssembler, SequenceAssembler


def _obs(t):
    return np.full((1, 2, 2, 1), t, dtype=np.uint8)


def test_nstep_reference_semantics():
    asm = NStepAssembler(1, 3, 0.5, reference_order=True, flush_tail=False)
    out = []
    rewards = [1, 2, 4, 8, 16]
    for t, r in enumerate(rewards):
        b = asm.step(_obs(t), [t], [r], _obs(t + 1), [t == 4])
        if b is not None:
            out.append(b)
    # windows [0,1,2],[1,2,3],[2,3,4]; reference weighting γ^{n-1-i}; only the last carries done (actor.py:78-81,90)
    assert [int(b.state[0, 0, 0, 0]) for b in out] == [0, 1, 2]
    assert [int(b.next_state[0, 0, 0, 0]) for b in out] == [3, 4, 5]
    assert np.allclose([b.reward[0] for b in out], [0.25 * 1 + 0.5 * 2 + 4, 0.25 * 2 + 0.5 * 4 + 8, 0.25 * 4 + 0.5 * 8 + 16])
    assert [b.done[0] for b in out] == [0, 0, 1]
    assert asm.step(_obs(9), [0], [1], _obs(10), [False]) is None          # buffer was cleared at done


def test_nstep_standard_with_tail_flush():
    asm = NStepAssembler(1, 3, 0.5)
    got = []
    for t, r in enumerate([1, 2, 4, 8]):
        b = asm.step(_obs(t), [t], [r], _obs(t + 1), [t == 3])
        if b is not None:
            got += list(zip(b.state[:, 0, 0, 0].tolist(), b.reward.tolist(), b.done.tolist(), b.next_state[:, 0, 0, 0].tolist()))
    assert got == [(0, 1 + 1 + 1, 0.0, 3), (1, 2 + 2 + 2, 1.0, 4), (2, 4 + 4, 1.0, 4), (3, 8.0, 1.0, 4)]


def test_nstep_vectorised_matches_single():
    rng = np.random.default_rng(0)
    E, T = 3, 40
    rew = rng.random((T, E))
    done = rng.random((T, E)) < 0.1
    vec = NStepAssembler(E, 3, 0.9)
    singles = [NStepAssembler(1, 3, 0.9) for _ in range(E)]
    tot_v = np.zeros(E)
    tot_s = np.zeros(E)
    for t in range(T):
        o = np.zeros((E, 1), np.uint8)
        b = vec.step(o, np.zeros(E, int), rew[t], o, done[t])
        if b is not None:
            np.add.at(tot_v, b.env, b.reward)
        for e in range(E):
            bs = singles[e].step(o[:1], [0], rew[t, e:e + 1], o[:1], done[t, e:e + 1])
            if bs is not None:
                tot_s[e] += bs.reward.sum()
    assert np.allclose(tot_v, tot_s)


def test_sequence_overlap_and_state():
    burn, seq, ov, n = 2, 4, 2, 2
    asm = SequenceAssembler(1, n, 1.0, burn, seq, ov, flush_tail=False)
    items = []
    for t in range(20):
        h = np.full((1, 3), t, np.float32)
        items += asm.step(_obs(t), [t], [1.0], [False], h, h)
    # total 6, keep burn+overlap = 4 → a new sequence every 2 steps; first completes when window t=5 closes (env step 6)
    starts = [int(s.frames[0, 0, 0, 0]) for s in items]
    assert starts[:4] == [0, 2, 4, 6]
    for s in items:
        assert s.length == 6
        assert s.h0[0] == s.frames[0, 0, 0, 0]                 # stored state = state before the first frame (r2d2/actor.py:39,57)
        assert np.all(np.diff(s.frames[:, 0, 0, 0].astype(int)) == 1)
        assert np.allclose(s.reward, 2.0)


def test_sequence_tail_on_done():
    burn, seq, ov, n = 2, 6, 2, 2
    asm = SequenceAssembler(1, n, 1.0, burn, seq, ov)          # flush_tail: terminal windows included
    out = []
    for t in range(9):
        h = np.zeros((1, 3), np.float32)
        out += asm.step(_obs(t), [0], [1.0], [t == 8], h, h)
    # 9 env steps → windows 0..7 full + 1 flushed tail (step 8) = 9 entries: one full sequence of 8, then the tail (4 kept + 1)
    assert out[0].length == 8
    # tail shorter than keep+n+1 is dropped (r2d2/actor.py:61), so only the full one is emitted
    assert len(out) == 1
    asm2 = SequenceAssembler(1, n, 1.0, burn, seq, ov)
    out2 = []
    for t in range(7):
        h = np.zeros((1, 3), np.float32)
        out2 += asm2.step(_obs(t), [0], [1.0], [t == 6], h, h)
    assert len(out2) == 1 and out2[0].length == 7 and out2[0].done[out2[0].length - 1] == 1.0
    assert np.all(out2[0].frames[7:] == 0)                     # zero padding past length


def test_catch_env_contract():
    env = envs.make_vec("MultiFrameBreakout-v0", 3, seed=0, backend="synthetic")
    assert env.n_action == 4
    o = env.reset()
    assert o.shape == (3, 84, 84, 4) and o.dtype == np.uint8
    total_done = 0
    for t in range(300):
        o2, r, d, info = env.step(np.random.randint(0, 4, 3))
        assert o2.shape == (3, 84, 84, 4) and r.shape == (3,) and d.shape == (3,)
        assert np.all(np.abs(r) <= 1)
        if d.any():
            total_done += int(d.sum())
            cur = env.current_obs()
            for e in np.nonzero(d)[0]:
                assert np.array_equal(o2[e], info["terminal_obs"][list(info["done_idx"]).index(e)])
                assert not np.array_equal(cur[e], o2[e])
    assert total_done > 0
    single = envs.make("MultiFramePong-v0", seed=1, backend="synthetic")
    assert single.action_space.n == 6
    s = single.reset()
    assert s.shape == (4, 84, 84) and s.dtype == np.float32 and s.max() <= 1.0     # wrapped_env.py:42 contract
    s2, r, d, _ = single.step(0)
    assert s2.shape == (4, 84, 84)
    assert len(envs.REGISTRY) >= 120 and "SingleFrameMsPacman-v0" in envs.REGISTRY


# ---- array-based assemblers (algo/vec_assembly.py) against the per-env oracle above ---------------------------------
def _episodes(rng, E, steps, p_done):
    for t in range(steps):
        yield (rng.integers(0, 256, (E, 6, 6, 2), dtype=np.uint8), rng.integers(0, 4, E), rng.standard_normal(E).astype(np.float32),
               rng.integers(0, 256, (E, 6, 6, 2), dtype=np.uint8), rng.random(E) < p_done)


@pytest.mark.parametrize("n,ref_order,flush", [(3, False, True), (3, True, False), (1, False, True), (5, False, True)])
def test_vec_nstep_assembler_equals_per_env_oracle(n, ref_order, flush):
    from distributed_rl_amd.algo.nstep import NStepAssembler
    from distributed_rl_amd.algo.vec_assembly import VecNStepAssembler
    E = 7
    a, b = NStepAssembler(E, n, 0.9, ref_order, flush), VecNStepAssembler(E, n, 0.9, ref_order, flush)
    rng = np.random.default_rng(n)
    emitted = 0
    for s, act, r, s2, d in _episodes(rng, E, 200, 0.08):
        oa, ob = a.step(s, act, r, s2, d), b.step(s, act, r, s2, d)
        assert (oa is None) == (ob is None)
        if oa is None:
            continue
        emitted += len(oa.env)
        key = lambda o: np.lexsort((o.reward, o.env))                 # noqa: E731 - same multiset, any order
        ka, kb = key(oa), key(ob)
        assert np.array_equal(oa.env[ka], ob.env[kb]) and np.array_equal(oa.action[ka], ob.action[kb])
        np.testing.assert_allclose(oa.reward[ka], ob.reward[kb], rtol=1e-6, atol=1e-6)
        assert np.array_equal(oa.done[ka], ob.done[kb])
        assert np.array_equal(oa.state[ka], ob.state[kb]) and np.array_equal(oa.next_state[ka], ob.next_state[kb])
    assert emitted > 500


@pytest.mark.parametrize("n,ref_order,flush,lazy", [(2, False, True, True), (2, True, False, False), (3, False, True, False)])
def test_vec_sequence_assembler_equals_per_env_oracle(n, ref_order, flush, lazy):
    from distributed_rl_amd.algo.nstep import SequenceAssembler
    from distributed_rl_amd.algo.vec_assembly import VecSequenceAssembler
    E, Hd = 5, 8
    kw = dict(burn_in=3, seq_len=6, overlap=2, reference_order=ref_order, flush_tail=flush)
    a, b = SequenceAssembler(E, n, 0.9, **kw), VecSequenceAssembler(E, n, 0.9, hidden=Hd, **kw)
    rng = np.random.default_rng(7)
    emitted, short, needed = 0, 0, 0
    for s, act, r, _, d in _episodes(rng, E, 400, 0.03):
        h, c = rng.standard_normal((E, Hd)).astype(np.float32), rng.standard_normal((E, Hd)).astype(np.float32)
        oa = a.step(s, act, r, d, h, c)
        need = b.need_state()
        needed += int(need.sum())
        ob = b.step(s, act, r, d, h[need], c[need]) if lazy else b.step(s, act, r, d, h, c)
        assert len(oa) == len(ob)
        sa = sorted(oa, key=lambda q: (q.length, float(q.reward.sum())))
        sb = sorted(ob, key=lambda q: (q.length, float(q.reward.sum())))
        for x, y in zip(sa, sb):
            emitted += 1
            short += x.length < a.total
            assert x.length == y.length and np.array_equal(x.frames, y.frames) and np.array_equal(x.action, y.action)
            np.testing.assert_allclose(x.reward, y.reward, rtol=1e-6, atol=1e-6)
            assert np.array_equal(x.done, y.done)
            assert np.array_equal(x.h0, y.h0) and np.array_equal(x.c0, y.c0)      # the state before the sequence's first frame
    assert emitted > 100 and (short > 3 or a.keep + n + 1 >= a.total)
    assert needed < 0.45 * 400 * E                                   # states are asked for at sequence starts only
