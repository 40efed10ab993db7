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


def test_async_ingest_apex_while_stepping(run_id):
    from distributed_rl_amd.engine.apex import ApexEngine
    from distributed_rl_amd.engine.ingest import AsyncIngest
    from distributed_rl_amd.replay.device_per import DevicePER
    dev = torch.device("cuda")
    cap, B = 300, 32
    rep = DevicePER(cap, dev, batch_size=B, seed=3)
    rep.fill_synthetic(100)
    eng = ApexEngine(4, ApexConfig(batch_size=B), rep, dev)
    eng.step()                                                       # warm-up + capture
    ring = ShmExperience(shm_names(run_id)[0], create=True, n_slots=8, slot_bytes=4 << 20)
    pipe = AsyncIngest(ring, rep, dev, depth=3)
    rng = np.random.default_rng(0)
    blocks = [_apex_block(rng, K) for K in (40, 64, 17, 64, 64, 50)]          # 299 items: wraps the 300-slot ring
    try:
        sent, start = 0, rep.cursor
        expect = {}
        pos = start
        for b in blocks:
            for i in range(b["action"].shape[0]):
                expect[(pos + i) % cap] = (b, i)
            pos += b["action"].shape[0]
        t0 = time.time()
        committed = 0
        while committed < 299 and time.time() - t0 < 60:
            if sent < len(blocks) and ring.push(blocks[sent], codec.KIND_APEX, 1, sent):
                sent += 1
            committed += pipe.service()
            eng.step()                                               # graph replays interleaved with the DMA
            # between retire and commit a slot has priority 0: whatever the sampler drew has a positive priority
            assert float(rep.prio.min()) > 0.0
        assert committed == 299 and pipe.pending() == 0
        torch.cuda.synchronize()
        eng.stats()
        assert len(rep) == cap and int(rep.n_items.item()) == cap     # count written by the commit launch
        fr = rep.frames.view(2, cap, 84, 84, 4)
        for slot in (start % cap, (start + 39) % cap, (start + 40) % cap, (start + 200) % cap, (start + 298) % cap):
            b, i = expect[slot]
            assert np.array_equal(fr[0, slot].cpu().numpy(), b["state"][i])
            assert np.array_equal(fr[1, slot].cpu().numpy(), b["next_state"][i])
            assert int(rep.act[slot]) == int(b["action"][i]) and float(rep.rew[slot]) == float(b["reward"][i])
            assert float(rep.done[slot]) == float(b["done"][i])
        # tree is exact: root == sum of leaves (leaves were partly rewritten by the engine's priority write-backs)
        leaves = rep.tree[:cap].double().sum().item()
        assert rep.total_prios() == pytest.approx(leaves, rel=1e-5)
        st = pipe.stats()
        assert st["items"] == 299 and st["blocks"] == 6 and st["bytes"] > 299 * 2 * 28224
    finally:
        pipe.close()
        ring.close()


def test_async_ingest_retired_slots_are_not_sampled_and_stale_writebacks_dropped(run_id):
    from distributed_rl_amd.engine.ingest import AsyncIngest
    from distributed_rl_amd.ops import hip as H
    from distributed_rl_amd.replay.device_per import DevicePER
    dev = torch.device("cuda")
    rep = DevicePER(64, dev, batch_size=64, seed=1)
    rep.fill_synthetic(64)
    st = H.new_step_state(dev, 0.4, 1e-6)
    gen0 = rep.gen.clone()
    rep.sample(st, begin=True)
    idx_before = rep.idx.clone()
    s = rep.reserve(32)                                               # slots 0..31 leave the tree
    rep.retire(s, 32)
    for _ in range(20):
        rep.sample(st, begin=True)
        assert int(rep.idx.min()) >= 32                              # never drawn while retired
    assert torch.equal(rep.gen[:32], gen0[:32] + 1) and torch.equal(rep.gen[32:], gen0[32:])
    # a write-back that sampled BEFORE the retire carries the old generation: dropped for the retired slots
    rep.idx.copy_(idx_before)
    rep.gen_s.copy_(gen0[idx_before.long()])
    rep.update(torch.full((64,), 123.0, device=dev))
    assert float(rep.tree[:32].abs().sum()) == 0.0
    rep.commit(s, torch.full((32,), 2.0, device=dev), 32)
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
    blk = dict(frames=rng.integers(0, 256, (K, T, 84, 84, 4), dtype=np.uint8), action=rng.integers(0, 4, (K, T)).astype(np.int32),
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
        assert np.array_equal(rep.frames.view(16, T, 84, 84, 4)[2].cpu().numpy(), blk["frames"][2])
        assert np.array_equal(rep.length[:K].cpu().numpy(), blk["length"])
        np.testing.assert_array_equal(rep.h0[:K].cpu().numpy(), blk["h0"])
        np.testing.assert_allclose(rep.tree[:K].cpu().numpy(), blk["prio"])
    finally:
        pipe.close()
