"""Asynchronous ingest (engine/ingest.py): shm ring -> pinned staging -> side-stream DMA -> retire / commit on the learner
stream, while the graph-captured engine keeps stepping. Checks the data that lands in HBM, the tree, the item count
(written by the commit launch) and that slots are invisible to the sampler between retire and commit."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from distributed_rl_amd.config import ApexConfig, R2D2Config  # noqa: E402
from distributed_rl_amd.transport import ShmExperience, codec, shm_names  # noqa: E402


def _apex_block(rng, K, lo=0.5):
    return dict(state=rng.integers(0, 256, (K, 84, 84, 4), dtype=np.uint8), next_state=rng.integers(0, 256, (K, 84, 84, 4), dtype=np.uint8),
                action=rng.integers(0, 4, K).astype(np.int32), reward=rng.standard_normal(K).astype(np.float32),
                done=(rng.random(K) < 0.1).astype(np.float32), prio=(rng.random(K) + lo).astype(np.float32))


def _plane_block(rng, runs, run, n=3):
    """A kind-3 block as the vectorised actor ships it: ``runs`` runs of ``run`` consecutive steps (s' = n steps on)."""
    per = run + n + 3
    planes = rng.integers(0, 256, (runs * per, 84, 84), dtype=np.uint8)
    i = np.arange(runs * run)
    ws = ((i // run) * per + i % run).astype(np.int32)
    K = len(i)
    return dict(planes=planes, win_s=ws, win_s2=(ws + n).astype(np.int32), action=rng.integers(0, 4, K).astype(np.int32),
                reward=rng.standard_normal(K).astype(np.float32), done=(rng.random(K) < 0.1).astype(np.float32),
                prio=(rng.random(K) + 0.5).astype(np.float32))


def _stack_of(planes, w):
    return np.ascontiguousarray(np.transpose(planes[w:w + 4], (1, 2, 0)))


def test_async_ingest_apex_while_stepping(run_id):
    """Blocks of both wire kinds (full stacks, plane pools) through shm ring -> pinned staging -> side-stream DMA while the
    graph-captured engine keeps stepping; enough of them to wrap both rings of the replay. Every transition still live
    afterwards reads back as the stacks the actor saw."""
    from distributed_rl_amd.engine.apex import ApexEngine
    from distributed_rl_amd.engine.ingest import AsyncIngest
    from distributed_rl_amd.replay.device_per import DevicePER
    dev = torch.device("cuda")
    cap, B = 300, 32
    rep = DevicePER(cap, dev, batch_size=B, seed=3, plane_capacity=1500)
    rep.fill_synthetic(100)
    eng = ApexEngine(4, ApexConfig(batch_size=B), rep, dev)
    eng.step()                                                       # warm-up + capture
    ring = ShmExperience(shm_names(run_id)[0], create=True, n_slots=8, slot_bytes=4 << 20)
    pipe = AsyncIngest(ring, rep, dev, depth=3)
    rng = np.random.default_rng(0)
    blocks = [(_apex_block(rng, 40), codec.KIND_APEX), (_plane_block(rng, 4, 16), codec.KIND_APEX_PLANES),
              (_apex_block(rng, 17), codec.KIND_APEX), (_plane_block(rng, 2, 32), codec.KIND_APEX_PLANES),
              (_plane_block(rng, 8, 8), codec.KIND_APEX_PLANES), (_apex_block(rng, 50), codec.KIND_APEX),
              (_plane_block(rng, 3, 20), codec.KIND_APEX_PLANES)]
    total = sum(b["action"].shape[0] for b, _ in blocks)              # 40 + 64 + 17 + 64 + 64 + 50 + 60 = 359 > 300 slots;
    try:                                                              # planes: 320 + 88 + 136 + 76 + 112 + 400 + 78
        sent, committed, t0 = 0, 0, time.time()
        while committed < total and time.time() - t0 < 60:
            if sent < len(blocks) and ring.push(blocks[sent][0], blocks[sent][1], 1, sent):
                sent += 1
            committed += pipe.service()
            eng.step()                                               # graph replays interleaved with the DMA
            # between retire and commit a slot has priority 0: whatever the sampler drew has a positive priority
            assert float(rep.prio.min()) > 0.0
        assert committed == total and pipe.pending() == 0
        torch.cuda.synchronize()
        eng.stats()
        live = sum(b.K for b in rep._blocks)
        assert len(rep) == live == int(rep.n_items.item()) and 0 < live <= cap     # count written by the commit launch
        # every block still owning its ranges reads back exactly; the blocks come back in ingestion order
        recs = list(rep._blocks)
        assert [r.K for r in recs[-3:]] == [64, 50, 60]
        for rec, (b, kind) in zip(recs[-3:], blocks[-3:]):
            slots = torch.arange(rec.slot, rec.slot + rec.K, device=dev)
            s_, s2_ = rep.gather_stacks(slots)
            for i in (0, rec.K // 2, rec.K - 1):
                if kind == codec.KIND_APEX:
                    es, es2 = b["state"][i], b["next_state"][i]
                else:
                    es, es2 = _stack_of(b["planes"], b["win_s"][i]), _stack_of(b["planes"], b["win_s2"][i])
                assert np.array_equal(s_[i].cpu().numpy(), es) and np.array_equal(s2_[i].cpu().numpy(), es2)
                assert int(rep.act[rec.slot + i]) == int(b["action"][i]) and float(rep.rew[rec.slot + i]) == float(b["reward"][i])
                assert float(rep.done[rec.slot + i]) == float(b["done"][i])
        # nothing outside the live blocks can be drawn: the leaves of every other slot are zero
        mask = torch.zeros(cap, dtype=torch.bool, device=dev)
        for r in recs:
            mask[r.slot:r.slot + r.K] = True
        assert float(rep.tree[:cap][~mask].abs().sum()) == 0.0 and float(rep.tree[:cap][mask].min()) > 0.0
        # tree is exact: root == sum of leaves (leaves were partly rewritten by the engine's priority write-backs)
        leaves = rep.tree[:cap].double().sum().item()
        assert rep.total_prios() == pytest.approx(leaves, rel=1e-5)
        st = pipe.stats()
        assert st["items"] == total and st["blocks"] == len(blocks)
    finally:
        pipe.close()
        ring.close()


def test_async_ingest_retired_slots_are_not_sampled_and_stale_writebacks_dropped(run_id):
    from distributed_rl_amd.ops import hip as H
    from distributed_rl_amd.replay.device_per import DevicePER
    dev = torch.device("cuda")
    rep = DevicePER(64, dev, batch_size=64, seed=1)
    rep.fill_synthetic(64, run=32)
    st = H.new_step_state(dev, 0.4, 1e-6)
    gen0 = rep.gen.clone()
    rep.sample(st, begin=True)
    idx_before = rep.idx.clone()
    rep._retire(0, 32)                                                # what reserve_block does to a range: slots 0..31 leave the tree
    for _ in range(20):
        rep.sample(st, begin=True)
        assert int(rep.idx.min()) >= 32                              # never drawn while retired
    assert torch.equal(rep.gen[:32], gen0[:32] + 1) and torch.equal(rep.gen[32:], gen0[32:])
    # a write-back that sampled BEFORE the retire carries the old generation: dropped for the retired slots
    rep.idx.copy_(idx_before)
    rep.gen_s.copy_(gen0[idx_before.long()])
    rep.update(torch.full((64,), 123.0, device=dev))
    assert float(rep.tree[:32].abs().sum()) == 0.0
    from distributed_rl_amd.replay.device_per import _Block
    rep.count = 32
    rep.commit_block(_Block(0, 32, 0, 36), torch.full((32,), 2.0, device=dev))
    assert float(rep.tree[:32].sum()) == 64.0 and int(rep.n_items.item()) == 64


def test_async_ingest_r2d2_inproc_channel(run_id):
    """Channels without zero-copy slots (in-process, Redis) go through the same pinned layout."""
    from distributed_rl_amd.engine.ingest import AsyncIngest
    from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
    from distributed_rl_amd.transport import connect
    dev = torch.device("cuda")
    cfg = R2D2Config(burn_in=2, seq_len=4, nstep=2, batch_size=4)
    rep = DeviceSequenceReplay(16, dev, cfg, seed=0)
    exp, _ = connect("inproc", run_id, create=True)
    pipe = AsyncIngest(exp, rep, dev, depth=2, slot_bytes=4 << 20)
    rng = np.random.default_rng(0)
    K, T = 5, cfg.total_len
    planes = rng.integers(0, 256, (K, T + 3, 84, 84), dtype=np.uint8)          # what an environment's frame stack is made of
    frames = np.stack([planes[:, k:k + T] for k in range(4)], -1)                # [K, T, 84, 84, 4]: observation t = planes t..t+3
    blk = dict(frames=frames, action=rng.integers(0, 4, (K, T)).astype(np.int32),
               reward=rng.standard_normal((K, T)).astype(np.float32), done=np.zeros((K, T), np.float32),
               length=np.array([T, T, T - 1, T, 3], np.int32), h0=rng.standard_normal((K, 512)).astype(np.float32),
               c0=rng.standard_normal((K, 512)).astype(np.float32), prio=(rng.random(K) + 0.5).astype(np.float32))
    try:
        assert exp.push(blk, codec.KIND_R2D2, 2, 0)
        t0 = time.time()
        n = 0
        while n < K and time.time() - t0 < 30:
            n += pipe.service()
            time.sleep(0.001)
        assert n == K and len(rep) == K and int(rep.n_items.item()) == K
        torch.cuda.synchronize()
        assert rep._blocks[-1].M == K * (T + 3) and rep.unshifted_blocks == 0                       # shift verified: compact form
        assert np.array_equal(rep.frames[2 * (T + 3):3 * (T + 3)].view(T + 3, 84, 84).cpu().numpy(), planes[2])   # every frame once
        assert np.array_equal(rep.gather_stacks(torch.tensor([2, 4], device=dev)).cpu().numpy(), blk["frames"][[2, 4]])
        assert np.array_equal(rep.length[:K].cpu().numpy(), blk["length"])
        np.testing.assert_array_equal(rep.h0[:K].cpu().numpy(), blk["h0"])
        np.testing.assert_allclose(rep.tree[:K].cpu().numpy(), blk["prio"])
    finally:
        pipe.close()


def test_async_ingest_sequence_plane_blocks_match_the_field_by_field_write(run_id):
    """Kind-4 blocks (sequences over a plane pool) through the pipeline's one-call write (two DMAs + the scatter launch,
    priorities through ``prio_pending``) against ``ingest_block`` (torch ops, field by
    field) into a second replay: every table, the planes, the tree and the item count; enough blocks to wrap the rings,
    several per service call so that retires and commits are coalesced."""
    from distributed_rl_amd.engine.ingest import AsyncIngest
    from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
    dev = torch.device("cuda")
    cfg = R2D2Config(burn_in=2, seq_len=4, nstep=2, batch_size=4)
    T = cfg.total_len
    rep = DeviceSequenceReplay(16, dev, cfg, seed=0, plane_capacity=200)
    ref = DeviceSequenceReplay(16, dev, cfg, seed=0, plane_capacity=200)
    ring = ShmExperience(shm_names(run_id)[0], create=True, n_slots=16, slot_bytes=2 << 20)
    pipe = AsyncIngest(ring, rep, dev, depth=8, slot_bytes=2 << 20)
    rng = np.random.default_rng(1)

    def block(K):
        per = T + 3
        return dict(planes=rng.integers(0, 256, (K * per, 84, 84), dtype=np.uint8),
                    win=(np.arange(K)[:, None] * per + np.arange(T)[None, :]).astype(np.int32),
                    action=rng.integers(0, 4, (K, T)).astype(np.int32), reward=rng.standard_normal((K, T)).astype(np.float32),
                    done=(rng.random((K, T)) < 0.1).astype(np.float32), length=rng.integers(3, T + 1, K).astype(np.int32),
                    h0=rng.standard_normal((K, 512)).astype(np.float32), c0=rng.standard_normal((K, 512)).astype(np.float32),
                    prio=(rng.random(K) + 0.5).astype(np.float32))
    blocks = [block(k) for k in (3, 2, 4, 1, 3, 2, 5, 3, 2)]          # 25 sequences into 16 slots / 225 planes into 200
    try:
        for b in blocks:
            assert ring.push(b, codec.KIND_R2D2_PLANES, 1, 0)
            ref.ingest_block(b)
        total, n, t0 = sum(b["length"].shape[0] for b in blocks), 0, time.time()
        while n < total and time.time() - t0 < 30:
            n += pipe.service()
            time.sleep(0.002)
        assert n == total and pipe.pending() == 0
        torch.cuda.synchronize()
        assert [(b.slot, b.K, b.plane, b.M, b.state) for b in rep._blocks] == [(b.slot, b.K, b.plane, b.M, b.state) for b in ref._blocks]
        assert len(rep) == len(ref) == int(rep.n_items.item()) == int(ref.n_items.item())
        live = torch.zeros(16, dtype=torch.bool, device=dev)
        for b in rep._blocks:
            if b.state == 1:
                live[b.slot:b.slot + b.K] = True
                assert torch.equal(rep.frames[b.plane:b.plane + b.M], ref.frames[b.plane:b.plane + b.M])
        assert int(live.sum()) == len(rep) > 0
        for name in ("win", "act", "rew", "done", "length", "h0", "c0"):
            assert torch.equal(getattr(rep, name)[live], getattr(ref, name)[live]), name
        assert torch.equal(rep.tree[:16], ref.tree[:16]) and float(rep.tree[:16][~live].abs().sum()) == 0.0
        assert rep.total_prios() == pytest.approx(ref.total_prios(), rel=1e-6)
    finally:
        pipe.close()
        ring.close()


def test_plane_store_against_the_host_oracle_through_episode_boundaries(run_id):
    """The actor's own plane-pool blocks (SyntheticCatch with one life: a reset every ~26 steps, truncated n-step tails onto
    terminal frames) into a replay small enough to lap both rings several times: every transition that is still live reads
    back - through the sampled-window path the kernels use - as exactly the stacks the host replay of the PyTorch learner
    would hold (runtime/replay_service.block_to_items), and nothing else can be drawn."""
    from distributed_rl_amd.envs.synthetic import CatchVecEnv
    from distributed_rl_amd.models import DuelingDQN
    from distributed_rl_amd.replay.device_per import DevicePER
    from distributed_rl_amd.runtime.actor import Actor
    from distributed_rl_amd.runtime.replay_service import block_to_items
    from distributed_rl_amd.transport import connect
    from distributed_rl_amd.config import make_config
    dev = torch.device("cuda")
    cfg = make_config("ape_x", actor_batch=40, nstep=3)
    exp, par = connect("inproc", run_id, create=True)
    E = 8
    actor = Actor(1, CatchVecEnv(E, seed=0, lives=1), DuelingDQN(4), cfg, exp, par, None, 1, torch.device("cpu"), seed=0)
    actor.run(max_steps=140)
    blocks = exp.drain(1 << 20)
    assert len(blocks) >= 20 and all(b["kind"] == codec.KIND_APEX_PLANES for b in blocks)
    assert sum(b["planes"].shape[0] for b in blocks) < 1.5 * sum(b["action"].shape[0] for b in blocks)    # vs 8 as stacks
    rep = DevicePER(300, dev, batch_size=16, seed=0, plane_capacity=700)
    owner = {}
    for b in blocks:
        rep.ingest_block(b)
        rec = rep._blocks[-1]
        items = block_to_items(b)
        for s in list(owner):                                          # whatever the replay evicted is gone from the oracle
            if not any(r.slot <= s < r.slot + r.K for r in rep._blocks if r.state == 1):
                del owner[s]
        for i, it in enumerate(items):
            owner[rec.slot + i] = (it, float(b["prio"][i]))
    torch.cuda.synchronize()
    live = sorted(owner)
    assert len(live) == len(rep) == int(rep.n_items.item()) and 100 < len(live) <= 300
    s_, s2_ = rep.gather_stacks(torch.tensor(live, device=dev))
    s_, s2_ = s_.cpu().numpy(), s2_.cpu().numpy()
    for j, slot in enumerate(live):
        (s, a, r, s2, d), p = owner[slot]
        assert np.array_equal(s_[j], s) and np.array_equal(s2_[j], s2)
        assert int(rep.act[slot]) == a and float(rep.rew[slot]) == pytest.approx(r) and float(rep.done[slot]) == d
        assert float(rep.tree[slot]) == pytest.approx(p)
    dead = torch.ones(300, dtype=torch.bool, device=dev)
    dead[torch.tensor(live, device=dev)] = False
    assert float(rep.tree[:300][dead].abs().sum()) == 0.0            # evicted / never written slots cannot be drawn
    # and the sampler hands the kernels exactly these windows
    from distributed_rl_amd.ops import hip as H
    st = H.new_step_state(dev, 0.4, 1e-6)
    rep.sample(st, begin=True)
    torch.cuda.synchronize()
    idx = rep.idx.long()
    assert torch.equal(rep.frame_off[:16], rep.win[idx]) and torch.equal(rep.frame_off[16:], rep.win[300 + idx])
    assert all(int(i) in owner for i in idx.cpu())


def test_sequence_plane_store_against_the_host_oracle_through_episode_boundaries_and_eviction(run_id):
    """The R2D2 actor's plane-pool blocks (codec kind 4: the sequences of ONE env with the planes they span once;
    SyntheticCatch with one life, so sequences end at resets and tails are shorter) into a replay small enough to lap both
    rings: every sequence that is still live reads back - through the window table the sampler uses - as exactly the
    stacks the host replay of the PyTorch learner would hold, planes shared by neighbouring sequences are evicted only
    with the whole block, and nothing evicted can be drawn."""
    from distributed_rl_amd.config import make_config
    from distributed_rl_amd.envs.synthetic import CatchVecEnv
    from distributed_rl_amd.models import DuelingLSTMDQN
    from distributed_rl_amd.ops import hip as H
    from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
    from distributed_rl_amd.runtime.actor import Actor
    from distributed_rl_amd.runtime.replay_service import block_to_items
    from distributed_rl_amd.transport import connect
    dev = torch.device("cuda")
    cfg = make_config("r2d2", burn_in=4, seq_len=8, overlap=4, nstep=2, actor_batch=4, batch_size=8)
    exp, par = connect("inproc", run_id, create=True)
    E, T = 6, cfg.total_len
    net, tgt = DuelingLSTMDQN(4, E, cfg.burn_in, cfg.nstep), DuelingLSTMDQN(4, E, cfg.burn_in, cfg.nstep)
    actor = Actor(1, CatchVecEnv(E, seed=0, lives=1), net, cfg, exp, par, None, 1, torch.device("cpu"), target_net=tgt, seed=0)
    actor.run(max_steps=260)
    blocks = exp.drain(1 << 20)
    assert len(blocks) >= 12 and all(b["kind"] == codec.KIND_R2D2_PLANES for b in blocks) and actor.lost_to_plane_log == 0
    n_seq, n_pl = sum(b["action"].shape[0] for b in blocks), sum(b["planes"].shape[0] for b in blocks)
    assert n_pl < 0.75 * n_seq * (T + 3)                            # sharing: fewer planes than sequences on their own
    rep = DeviceSequenceReplay(40, dev, cfg, seed=0, plane_capacity=260)
    owner = {}
    for b in blocks:
        rep.ingest_block(b)
        rec = rep._blocks[-1]
        for s in list(owner):                                       # whatever the replay evicted is gone from the oracle
            if not any(r.slot <= s < r.slot + r.K for r in rep._blocks if r.state == 1):
                del owner[s]
        for i, it in enumerate(block_to_items(b)):
            owner[rec.slot + i] = (it, float(b["prio"][i]))
    torch.cuda.synchronize()
    live = sorted(owner)
    assert rep.evicted_items > 0 and len(live) == len(rep) == int(rep.n_items.item()) and 8 <= len(live) <= 40
    got = rep.gather_stacks(torch.tensor(live, device=dev)).cpu().numpy()
    for j, slot in enumerate(live):
        it, p = owner[slot]
        L = it["length"]
        assert int(rep.length[slot]) == L and np.array_equal(got[j][:L], it["frames"][:L])
        assert np.array_equal(rep.act[slot].cpu().numpy()[:L], it["action"][:L])
        np.testing.assert_allclose(rep.rew[slot].cpu().numpy()[:L], it["reward"][:L])
        np.testing.assert_array_equal(rep.h0[slot].cpu().numpy(), it["h0"])
        assert float(rep.tree[slot]) == pytest.approx(p)
    dead = torch.ones(40, dtype=torch.bool, device=dev)
    dead[torch.tensor(live, device=dev)] = False
    assert float(rep.tree[:40][dead].abs().sum()) == 0.0           # evicted / never written slots cannot be drawn
    st = H.new_step_state(dev, 0.6, 1e-6)
    rep.sample(st, begin=True)
    torch.cuda.synchronize()
    idx = rep.idx.long()
    assert all(int(i) in owner for i in idx.cpu())
    assert torch.equal(rep.frame_off.view(T, cfg.batch_size), rep.win[idx].t())     # time-major window offsets of the batch


def test_sequence_block_of_stacks_that_do_not_shift_is_stored_uncompressed_and_exactly():
    """full-stack sequence blocks (codec kind 2) from an env whose stack does not move by one frame per
    step used to be split as if it did. The pixels are checked now: such a block keeps all four planes of every step."""
    from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
    dev = torch.device("cuda")
    cfg = R2D2Config(burn_in=2, seq_len=4, nstep=2, batch_size=4)
    rep = DeviceSequenceReplay(8, dev, cfg, seed=0)
    rng = np.random.default_rng(3)
    K, T = 3, cfg.total_len
    frames = rng.integers(0, 256, (K, T, 84, 84, 4), dtype=np.uint8)                 # unrelated stacks
    blk = dict(frames=frames, action=rng.integers(0, 4, (K, T)).astype(np.int32), reward=np.zeros((K, T), np.float32),
               done=np.zeros((K, T), np.float32), length=np.array([T, T, 2], np.int32), h0=np.zeros((K, 512), np.float32),
               c0=np.zeros((K, 512), np.float32), prio=np.ones(K, np.float32))
    rep.ingest_block(blk)
    torch.cuda.synchronize()
    assert rep.unshifted_blocks == 1 and rep._blocks[-1].M == K * 4 * T
    assert np.array_equal(rep.gather_stacks(torch.arange(K, device=dev)).cpu().numpy(), frames)
