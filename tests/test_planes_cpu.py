"""Every frame once: the actor-side plane stream (algo/vec_assembly.PlaneLog), the positions the n-step
assembler attaches to its transitions and the plane-pool wire block (codec kind 3), against the full stacks the same
assembler emits - episode starts (reset stack = 4 copies + no-ops), episode tails (truncated n-step windows onto the
terminal frame), environments that do not stack at all, a log that is too small."""
import numpy as np
import pytest

from distributed_rl_amd.algo.vec_assembly import PlaneLog, VecNStepAssembler, plane_block
from distributed_rl_amd.envs.synthetic import CatchVecEnv, RandomFrameVecEnv
from distributed_rl_amd.runtime.replay_service import block_to_items
from distributed_rl_amd.transport import codec


def _stack(planes, w):
    return np.ascontiguousarray(np.transpose(planes[w:w + 4], (1, 2, 0)))


def _drive(env, E, n, steps, cap=64, seed=0):
    asm = VecNStepAssembler(E, n, 0.99)
    log = PlaneLog(E, cap=cap)
    env.reset()
    rng = np.random.default_rng(seed)
    batches, episodes = [], 0
    for _ in range(steps):                                        # the actor's loop (runtime/actor.py)
        obs = env.current_obs()
        qs = log.observe(obs)
        act = rng.integers(0, 4, E)
        nxt, r, d, _ = env.step(act)
        qn = log.observe(nxt)
        out = asm.step(obs, act, r, nxt, d, qs, qn)
        if out is not None:
            planes, ws, ws2 = plane_block(log, out.env, out.qs, out.qs2)
            batches.append((out, planes, ws, ws2))
        episodes += int(d.sum())
    return batches, episodes, log


def test_plane_positions_reproduce_every_stack_across_episode_boundaries():
    E, n = 6, 3
    batches, episodes, log = _drive(CatchVecEnv(E, seed=0, lives=1), E, n, 300)
    assert episodes > 30                                          # many resets and truncated tails in the run
    tails = total = 0
    for out, planes, ws, ws2 in batches:
        for i in range(len(out.env)):
            assert np.array_equal(_stack(planes, ws[i]), out.state[i])
            assert np.array_equal(_stack(planes, ws2[i]), out.next_state[i])
        # inside an episode s' is exactly n planes on; only the truncated windows at an episode's end are closer
        gap = out.qs2 - out.qs
        assert gap.max() <= n and gap.min() >= 1 and np.all(gap[out.done == 0] == n)
        tails += int((gap < n).sum())
        total += len(gap)
    assert tails > 30 and total > 1500
    assert log.head.sum() / (300 * E) < 1.25                      # ~1 plane per env step + 4 per reset (full stacks: 8)


def test_an_env_that_does_not_stack_is_stored_exactly_just_not_compressed():
    E, n = 3, 2
    batches, _, log = _drive(RandomFrameVecEnv(E, 4, episode_len=17, seed=1), E, n, 60, cap=256)
    for out, planes, ws, ws2 in batches:
        for i in range(len(out.env)):
            assert np.array_equal(_stack(planes, ws[i]), out.state[i])
            assert np.array_equal(_stack(planes, ws2[i]), out.next_state[i])
    assert log.head.sum() >= 4 * 2 * 60 * E                       # every observation (s and s' differ here) entered whole


def test_assembler_without_state_copies_emits_the_same_positions():
    """store_states=False (the actor's setting once a PlaneLog keeps the frames): identical actions / returns / dones / envs /
    positions, no observation copied into the batches."""
    E, n = 5, 3
    env = CatchVecEnv(E, seed=4, lives=1)
    a1, a2 = VecNStepAssembler(E, n, 0.99), VecNStepAssembler(E, n, 0.99, store_states=False)
    log = PlaneLog(E, cap=64)
    env.reset()
    rng = np.random.default_rng(1)
    emitted = 0
    for _ in range(200):
        obs = env.current_obs()
        qs = log.observe(obs)
        act = rng.integers(0, 4, E)
        nxt, r, d, _ = env.step(act)
        qn = log.observe(nxt)
        o1, o2 = a1.step(obs, act, r, nxt, d, qs, qn), a2.step(obs, act, r, nxt, d, qs, qn)
        assert (o1 is None) == (o2 is None)
        if o1 is None:
            continue
        assert o2.state is None and o2.next_state is None
        for f in ("action", "reward", "done", "env", "qs", "qs2"):
            assert np.array_equal(getattr(o1, f), getattr(o2, f)), f
        emitted += len(o1.env)
    assert emitted > 800
    with pytest.raises(AssertionError):
        a2.step(obs, act, r, nxt, d)                              # no positions, no states: nothing to emit from


def test_plane_log_refuses_positions_it_no_longer_holds():
    log = PlaneLog(1, cap=8)
    obs = np.zeros((1, 84, 84, 4), np.uint8)
    log.observe(obs)
    for k in range(1, 12):
        nxt = np.ascontiguousarray(np.concatenate([obs[..., 1:], np.full((1, 84, 84, 1), k, np.uint8)], -1))
        log.observe(nxt)
        assert int(log.observe(nxt)[0]) == k                      # seeing it again changes nothing
        obs = nxt
    assert int(log.q[0]) == 11 and int(log.head[0]) == 15
    assert np.array_equal(log.planes(0, 11, 14)[:, 0, 0], [8, 9, 10, 11])
    with pytest.raises(IndexError):
        log.planes(0, 3, 6)                                       # overwritten
    with pytest.raises(IndexError):
        log.planes(0, 12, 15)                                     # not there yet


def test_plane_block_wire_format_roundtrip_and_host_replay_items():
    E, n = 4, 3
    batches, _, log = _drive(CatchVecEnv(E, seed=3, lives=2), E, n, 16)
    from distributed_rl_amd.algo.nstep import NStepBatch
    out = NStepBatch(*[np.concatenate([getattr(b[0], f) for b in batches]) for f in NStepBatch._fields])     # one actor block
    planes, ws, ws2 = plane_block(log, out.env, out.qs, out.qs2)
    K = len(out.env)
    assert K >= 48
    blk = dict(planes=planes, win_s=ws, win_s2=ws2, action=out.action.astype(np.int32), reward=out.reward, done=out.done,
               prio=np.linspace(0.1, 1.0, K).astype(np.float32))
    buf = codec.encode(blk, codec.KIND_APEX_PLANES, actor_id=3, env_steps=99)
    assert len(buf) == codec.apex_planes_block_bytes(K, planes.shape[0], (84, 84, 4)) < codec.apex_block_bytes(K, (84, 84, 4)) // 3
    meta, fields = codec.layout(buf)
    assert meta["kind"] == codec.KIND_APEX_PLANES and meta["K"] == K and meta["M"] == planes.shape[0] and meta["bytes"] == len(buf)
    dec = codec.decode(buf)
    for k in blk:
        assert np.array_equal(dec[k], blk[k]), k
    items = block_to_items(dec)                                   # the host PER (PyTorch learner path) gets full stacks back
    for i, (s, a, r, s2, d) in enumerate(items):
        assert np.array_equal(s, out.state[i]) and np.array_equal(s2, out.next_state[i])
        assert a == int(out.action[i]) and r == pytest.approx(float(out.reward[i])) and d == float(out.done[i])


def test_actor_with_a_non_shifting_env_publishes_short_blocks_instead_of_losing_them(run_id):
    """``SyntheticRandom-v0`` restarts the stack on every observation (8 planes per env step), so a
    default ``actor_batch`` of 50 transitions would span more planes than the actor's PlaneLog holds. The actor publishes
    such an env early, as shorter blocks; nothing is dropped and every published window is the stack the env showed."""
    from distributed_rl_amd import envs
    from distributed_rl_amd.config import make_config
    from distributed_rl_amd.models import DuelingDQN
    from distributed_rl_amd.runtime.actor import Actor
    from distributed_rl_amd.transport import connect
    cfg = make_config("ape_x")                                    # defaults: actor_batch 50, nstep 3
    exp, par = connect("inproc", run_id, create=True)
    env = envs.make_vec("SyntheticRandom-v0", 2, seed=0)
    shown = []
    step0 = env.step

    def spy(a):                                                   # record every observation the env ever delivers
        out = step0(a)
        shown.append(out[0].copy())
        return out
    env.step = spy
    actor = Actor(1, env, DuelingDQN(env.n_action), cfg, exp, par, None, 1, seed=0)
    actor.run(max_steps=120)
    assert actor.published_blocks > 0 and actor.dropped_blocks == 0 and actor.lost_to_plane_log == 0
    blocks = exp.drain()
    sizes = [int(b["action"].shape[0]) for b in blocks]
    assert sum(sizes) >= 2 * (120 - cfg.actor_batch) and max(sizes) < cfg.actor_batch          # short blocks, nearly every step
    seen = {o[e].tobytes() for o in shown for e in range(2)}
    for b in blocks:
        for w in b["win_s2"]:                                     # s' of every transition is a stack the env really showed
            assert np.ascontiguousarray(np.transpose(b["planes"][w:w + 4], (1, 2, 0))).tobytes() in seen


def test_native_planelog_observe_equals_the_numpy_form():
    """csrc/host/drl_host.cpp drl_planelog_observe (one pass per env: relation, new planes into the ring, copy of the stack)
    against PlaneLog's numpy form on the same stream of observations: unchanged stacks, stacks moved on by one frame, resets
    (four copies of a frame), unrelated stacks, a stack that differs from a shifted one in ONE pixel of an old plane, and
    enough steps to wrap the plane ring."""
    from distributed_rl_amd.algo import vec_assembly as va
    if va._native_observe is None:
        pytest.skip("host library not built")
    E, H, W = 5, 12, 10
    rng = np.random.default_rng(3)
    a, b = va.PlaneLog(E, cap=24, native=True), va.PlaneLog(E, cap=24, native=False)
    assert a.native and not b.native
    cur = rng.integers(0, 256, (E, H, W, 4), dtype=np.uint8)
    kinds = set()
    for t in range(300):
        nxt = cur.copy()
        for e in range(E):
            k = rng.integers(0, 10)
            if k < 6:                                                    # one new frame
                nxt[e] = np.concatenate([cur[e][..., 1:], rng.integers(0, 256, (H, W, 1), dtype=np.uint8)], -1)
            elif k == 6:                                                 # reset: four copies of a first frame
                nxt[e] = np.repeat(rng.integers(0, 256, (H, W, 1), dtype=np.uint8), 4, -1)
            elif k == 7:                                                 # unrelated
                nxt[e] = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            elif k == 8:                                                 # almost a shift: one pixel of an OLD plane differs
                nxt[e] = np.concatenate([cur[e][..., 1:], rng.integers(0, 256, (H, W, 1), dtype=np.uint8)], -1)
                nxt[e][rng.integers(0, H), rng.integers(0, W), rng.integers(0, 3)] ^= 1 << int(rng.integers(0, 8))
            kinds.add(int(k))
        cur = np.ascontiguousarray(nxt)
        qa, qb = a.observe(cur), b.observe(cur)
        assert np.array_equal(qa, qb) and np.array_equal(a.head, b.head) and np.array_equal(a.q, b.q), t
        assert np.array_equal(a._last, b._last) and np.array_equal(a.P, b.P), t
    assert len(kinds) == 10 and int(a.head.min()) > 10 * a.cap           # every case seen, the ring lapped
    for e in range(E):                                                   # and the current stack reads back from the ring
        assert np.array_equal(np.moveaxis(a.planes(e, int(a.q[e]), int(a.q[e]) + 3), 0, -1), cur[e])
