"""Shared-memory experience ring / param register / heartbeats with real processes,
plus codec round trips."""
import multiprocessing as mp
import time

import numpy as np
import pytest

from distributed_rl_amd.transport import ShmExperience, ShmParams, codec, connect


def _apex_block(K, seed):
    rng = np.random.default_rng(seed)
    return dict(state=rng.integers(0, 256, (K, 84, 84, 4), dtype=np.uint8),
                next_state=rng.integers(0, 256, (K, 84, 84, 4), dtype=np.uint8),
                action=rng.integers(0, 4, K).astype(np.int32), reward=rng.random(K).astype(np.float32),
                done=(rng.random(K) < 0.1).astype(np.float32), prio=rng.random(K).astype(np.float32))


def test_codec_round_trip():
    b = _apex_block(7, 0)
    raw = codec.encode(b, codec.KIND_APEX, actor_id=3, env_steps=99)
    d = codec.decode(raw)
    assert d["kind"] == codec.KIND_APEX and d["actor_id"] == 3 and d["env_steps"] == 99
    for k, v in b.items():
        assert np.array_equal(d[k], v)
    rng = np.random.default_rng(1)
    s = dict(frames=rng.integers(0, 256, (2, 5, 84, 84, 4), dtype=np.uint8), action=rng.integers(0, 4, (2, 5)).astype(np.int32),
             reward=rng.random((2, 5)).astype(np.float32), done=np.zeros((2, 5), np.float32), length=np.array([5, 3], np.int32),
             h0=rng.random((2, 512)).astype(np.float32), c0=rng.random((2, 512)).astype(np.float32), prio=rng.random(2).astype(np.float32))
    d = codec.decode(codec.encode(s, codec.KIND_R2D2))
    for k, v in s.items():
        assert np.array_equal(d[k], v)


def _producer(name, pid, n_blocks):
    ch = ShmExperience(name)
    sent = 0
    while sent < n_blocks:
        blk = _apex_block(4, pid * 1000 + sent)
        blk["reward"][:] = pid * 1000 + sent
        if ch.push(blk, codec.KIND_APEX, actor_id=pid):
            sent += 1
        else:
            time.sleep(0.001)
    ch.close()


def test_shm_ring_multi_producer(run_id):
    name = "/drl_%s_exp" % run_id
    ring = ShmExperience(name, create=True, n_slots=4, slot_bytes=codec.apex_block_bytes(4, (84, 84, 4)))
    n_prod, per = 3, 12
    procs = [mp.Process(target=_producer, args=(name, p, per)) for p in range(n_prod)]
    for p in procs:
        p.start()
    seen = []
    t0 = time.time()
    while len(seen) < n_prod * per and time.time() - t0 < 60:
        for b in ring.drain():
            tag = int(b["reward"][0])
            ref = _apex_block(4, tag)
            assert np.array_equal(b["state"], ref["state"]) and b["actor_id"] == tag // 1000
            seen.append(tag)
    for p in procs:
        p.join(10)
    assert sorted(seen) == sorted(p * 1000 + i for p in range(n_prod) for i in range(per))   # at-most-once, nothing lost
    for p in range(n_prod):                                                                  # per-producer FIFO
        mine = [s for s in seen if s // 1000 == p]
        assert mine == sorted(mine)
    st = ring.stats()
    assert st["pushed"] == n_prod * per and st["size"] == 0
    ring.close()


def test_shm_ring_full_reset_oversize(run_id):
    name = "/drl_%s_exp" % run_id
    ring = ShmExperience(name, create=True, n_slots=2, slot_bytes=codec.apex_block_bytes(2, (84, 84, 4)))
    b = _apex_block(2, 0)
    assert ring.push(b, codec.KIND_APEX) and ring.push(b, codec.KIND_APEX)
    assert not ring.push(b, codec.KIND_APEX)            # bounded: back-pressure instead of unbounded growth
    assert ring.size() == 2
    ring.reset()                                          # DEL experience
    assert ring.size() == 0 and ring.drain() == []
    with pytest.raises(ValueError):
        ring.push(_apex_block(3, 0), codec.KIND_APEX)
    ring.close()


def _param_reader(name, q):
    par = ShmParams(name)
    have, seen, bad = 0, [], 0
    t0 = time.time()
    while time.time() - t0 < 20:
        got = par.pull(have)
        if got is None:
            continue
        have, flat = got
        seen.append(have)
        if not np.all(flat == flat[0]) or int(flat[0]) != have:
            bad += 1
        if have >= 200:
            break
    q.put((seen, bad))


def test_param_register_latest_value(run_id):
    name = "/drl_%s_par" % run_id
    par = ShmParams(name, create=True, capacity=1 << 20)
    assert par.pull(0) is None and par.version() == 0     # nothing before the first publish (actor.py:53-54)
    q = mp.Queue()
    p = mp.Process(target=_param_reader, args=(name, q))
    p.start()
    for v in range(1, 201):
        assert par.publish(np.full(200_000, v, dtype=np.float32)) == v
        if v % 20 == 0:
            time.sleep(0.002)
    seen, bad = q.get(timeout=30)
    p.join(10)
    assert bad == 0                                        # never a torn read
    assert seen == sorted(set(seen)) and seen[-1] == 200   # monotone, may skip versions
    par.reset()
    assert par.pull(0) is None
    with pytest.raises(ValueError):
        par.publish(np.zeros(1 << 20, dtype=np.float32))
    par.close()


def test_heartbeats(run_id):
    from distributed_rl_amd.runtime.heartbeat import HeartbeatTable
    tab = HeartbeatTable("/drl_%s_hb" % run_id, create=True)
    a = tab.register("actor", 3)
    b = HeartbeatTable("/drl_%s_hb" % run_id).register("learner", 0)
    a.tick(17)
    b.tick(5)
    snap = {(r["role"], r["ident"]): r for r in tab.snapshot()}
    assert snap[("actor", 3)]["progress"] == 17 and snap[("learner", 0)]["beats"] == 1
    assert snap[("actor", 3)]["age_s"] < 5
    a.close()
    assert ("actor", 3) not in {(r["role"], r["ident"]) for r in tab.snapshot()}
    tab.close()


def test_inproc_transport():
    exp, par = connect("inproc", "x", create=True)
    exp2, par2 = connect("inproc", "x")
    assert exp is exp2
    exp.push(_apex_block(2, 0), codec.KIND_APEX)
    assert exp.size() == 1 and len(exp2.drain()) == 1
    assert par2.pull(0) is None
    par.publish(np.arange(4, dtype=np.float32))
    v, flat = par2.pull(0)
    assert v == 1 and flat.tolist() == [0, 1, 2, 3] and par2.pull(1) is None
