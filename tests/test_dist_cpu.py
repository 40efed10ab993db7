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
        v, t = ch.pull_tensor(1)                                  # device-side hand-over: a private tensor, no numpy copy
        assert v == 2 and isinstance(t, torch.Tensor) and t.device == ch.device and float(t[0]) == 7.0 and ch.pull_tensor(2) is None
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


# ---- sharded PER (M8): IS weights of per-shard sampling vs the single-tree oracle ------------------------------------
def _shard_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from distributed_rl_amd.engine.shard import ShardedIS
    from distributed_rl_amd.ops import oracle
    from distributed_rl_amd.parallel.dist import Dist
    dist = Dist(device=torch.device("cpu"))

    class FakeReplay:
        n_items = torch.zeros(1, dtype=torch.int64)

    class Eng(ShardedIS):
        steps_done = 0

    # unequal shards: rank r holds 100 * (r + 1) + 1 items (odd total: the rounding of N_global / world is exercised)
    g = torch.Generator().manual_seed(5)
    all_p = [torch.rand(100 * (r + 1) + 1, generator=g) + 0.05 for r in range(world)]
    # scale every shard to the same Σp: then per-shard sampling draws item i with p_i / Σp_global, exactly like ONE tree
    all_p = [p * (50.0 / float(p.sum())) for p in all_p]
    mine = all_p[rank]
    e = Eng()
    e.replay, e.dist = FakeReplay(), dist
    e.replay.n_items = torch.tensor([mine.numel()])
    e._init_shard(e.replay, dist, torch.device("cpu"))
    assert e._shard_on and e.n_is is not e.replay.n_items
    e._maybe_refresh_shard()                                   # first step: collective
    n_global = sum(p.numel() for p in all_p)
    assert e.shard_n_is() == (n_global + world // 2) // world
    beta = 0.6
    pick = torch.arange(0, mine.numel(), 7)
    # what the kernels compute: (n_is * p / Σp_local)^-β (normalisation left out: it is per batch, see engine/shard.py)
    got = (float(e.shard_n_is()) * mine[pick].double() / float(mine.sum())).pow(-beta)
    single = (float(n_global) * mine[pick].double() / float(sum(p.sum() for p in all_p))).pow(-beta)
    torch.testing.assert_close(got, single, rtol=2e-3, atol=0)          # rtol: N_global / world is rounded to an integer
    w = oracle.sharded_is_weights(mine[pick], mine.sum(), n_global, world, beta)
    torch.testing.assert_close(w.double(), single / single.max(), rtol=1e-4, atol=0)
    # shards grow at different rates: the refresh (every shard_refresh steps, outside the graph) follows
    e.replay.n_items += 1000 * (rank + 1)
    e.steps_done = e.shard_refresh
    e._maybe_refresh_shard()
    n_global += 1000 * sum(range(1, world + 1))
    assert e.shard_n_is() == (n_global + world // 2) // world
    e.steps_done = e.shard_refresh + 1
    e.replay.n_items += 5
    e._maybe_refresh_shard()                                   # not due: unchanged
    assert e.shard_n_is() == (n_global + world // 2) // world
    # unequal Σp as well: the weights keep the estimator unbiased for the distribution actually sampled (β = 1):
    # Σ_shards Σ_i P(i) w_i f_i = mean f, with P(i) = p_i / (world Σp_shard), w_i = 1 / (N_global P(i))
    p2 = mine * (rank + 1.0)
    f = torch.rand(p2.numel(), generator=torch.Generator().manual_seed(rank)).double()
    P = p2.double() / (world * p2.double().sum())
    w1 = (n_global / world * p2.double() / p2.double().sum()).pow(-1.0)
    part = torch.tensor([float((P * w1 * f).sum()), float(f.sum())], dtype=torch.float64)
    dist.all_reduce_flat(part)
    assert abs(float(part[0]) - float(part[1]) / n_global) < 1e-9
    dist.barrier()
    dist.close()
    open(os.path.join(out_dir, "shard_ok%d" % rank), "w").close()


def test_sharded_is_weights_equal_single_tree_oracle(tmp_path):
    world = 2
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("shard_ok%d" % r)) for r in range(world))


# ---- bench.py launches its own ranks (the form the driver uses: python bench.py --gpus N ...) ----------------------
def _bench(args, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_bench_self_launches_ranks_and_relays_one_line():
    import json
    r = _bench(["--gpus", "2", "--check-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line on stdout, everything else went to stderr
    res = json.loads(lines[0])
    assert res["value"] == 2 and res["n_gpus"] == 2


def test_bench_fails_when_a_rank_fails():
    r = _bench(["--gpus", "2", "--check-ranks"], {"DRL_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0 and not r.stdout.strip()
