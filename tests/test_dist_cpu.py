"""Data-parallel correctness without a cluster: real ``torch.distributed`` ranks over gloo on
CPU, world_size 2. Gradient averaging equivalence (1×B vs 2×B/2), flat all-reduce / broadcast, shard scalars."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from distributed_rl_amd.models import DuelingDQN, ParamPack
    from distributed_rl_amd.parallel.dist import Dist
    dist = Dist(device=torch.device("cpu"))
    assert dist.backend == "gloo" and dist.enabled and dist.world_size == world
    torch.manual_seed(0)                                   # same init + same global batch on every rank
    model = DuelingDQN(4)
    target = DuelingDQN(4)
    g = torch.Generator().manual_seed(1)
    B = 8
    s = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, generator=g)
    s2 = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, generator=g)
    a = torch.randint(0, 4, (B,), generator=g)
    r, d, w = torch.randn(B, generator=g), torch.zeros(B), torch.rand(B, generator=g)
    # full-batch reference gradient
    model.td_batch(target, s, a, r, s2, d, w, 0.6, 0.99).loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    model.zero_grad()
    lo, hi = rank * B // world, (rank + 1) * B // world
    model.td_batch(target, s[lo:hi], a[lo:hi], r[lo:hi], s2[lo:hi], d[lo:hi], w[lo:hi], 0.6, 0.99).loss.backward()
    dist.all_reduce_grads(model)
    got = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-6)
    # flat-bucket path used by the engines: SUM then 1/world inside the optimizer
    pack = ParamPack.from_module(model)
    flat = torch.full_like(pack.flat, float(rank + 1))
    dist.all_reduce_flat(flat)
    assert float(flat[0]) == sum(range(1, world + 1))
    p0 = torch.full((16,), float(rank))
    dist.broadcast_flat(p0, src=0)
    assert float(p0.sum()) == 0.0
    tot, n = dist.shard_scalars(10.0 * (rank + 1), 100 * (rank + 1))
    assert tot == 10.0 * sum(range(1, world + 1)) and n == 100 * sum(range(1, world + 1))
    assert dist.max_scalar(float(rank)) == world - 1
    dist.barrier()
    # parameter sync by broadcast (RCCL on GPUs, gloo here): rank 0 = learner, rank 1 = an actor with a receiver thread
    from distributed_rl_amd.transport.collective_params import CollectiveParams
    import time
    ch = CollectiveParams(pack.flat.numel(), src_rank=0)
    if rank == 0:
        assert ch.publish(torch.full_like(pack.flat, 3.0)) == 1
        assert ch.publish(pack.flat.numpy() * 0 + 7.0) == 2
        ch.close()
    else:
        seen, t0 = 0, time.time()
        while seen < 2 and time.time() - t0 < 60:
            got = ch.pull(seen)
            if got is None:
                time.sleep(0.01)
                continue
            seen, flat = got
            assert flat.shape == (pack.flat.numel(),) and float(flat[0]) == (3.0 if seen == 1 else 7.0) and float(flat[-1]) == float(flat[0])
        assert seen == 2 and ch.pull(2) is None
        ch.close()
    dist.barrier()
    dist.close()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_torch_learner_with_dist_single_rank(run_id):
    """world_size=1 degenerate path: Dist is a no-op and the learner step is unchanged."""
    from distributed_rl_amd.parallel.dist import Dist
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    d = Dist(device=torch.device("cpu"))
    assert not d.enabled
    t = torch.ones(4)
    d.all_reduce_flat(t)
    d.broadcast_flat(t)
    d.barrier()
    assert d.shard_scalars(2.0, 3) == (2.0, 3) and d.max_scalar(1.5) == 1.5
