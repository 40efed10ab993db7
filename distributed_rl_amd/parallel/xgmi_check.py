"""Self-check / micro-benchmark of the direct xGMI all-reduce; one process per rank:

    RANK=r WORLD_SIZE=W MASTER_ADDR=127.0.0.1 MASTER_PORT=p python -m distributed_rl_amd.parallel.xgmi_check [--numel N]

Ranks share GPUs round-robin when there are fewer GPUs than ranks (a 1-GPU box runs W ranks on one device: the IPC
mapping, the flag protocol and graph replay are exercised, the links are not). The handles travel over a gloo group, the
expected sums are computed on the host. Prints one JSON line per rank; exit code 0 = every comparison passed."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as td


def engine_check(r2d2: bool) -> int:
    """W data-parallel learners (one process each) whose step graph contains the direct all-reduce: parameters must stay
    bit-identical across ranks (every slice is reduced by exactly one rank and copied by the others)."""
    from distributed_rl_amd.parallel.dist import Dist
    os.environ.setdefault("DRL_DIST_BACKEND", "gloo")
    os.environ.setdefault("DRL_SHARE_GPU", "1")
    os.environ.setdefault("LOCAL_RANK", os.environ["RANK"])
    os.environ["DRL_XGMI_AR"] = "1"                               # required: a failure raises instead of falling back
    dist = Dist()
    torch.manual_seed(0)                                          # same initial weights on every rank
    if r2d2:
        from distributed_rl_amd.config import R2D2Config
        from distributed_rl_amd.engine.r2d2 import R2D2Engine
        from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
        # ranks share one GPU here: two grid-wide persistent recurrences cannot be co-resident, use the per-step launches
        os.environ.setdefault("DRL_PERSISTENT_LSTM", "0")
        cfg = R2D2Config(batch_size=8, burn_in=4, seq_len=8, nstep=2, target_update=10 ** 9)
        rep = DeviceSequenceReplay(40, dist.device, cfg, seed=5 + dist.rank)
        rep.fill_synthetic(40, n_action=4, seed=dist.rank)         # every rank owns a different shard
        eng = R2D2Engine(4, cfg, rep, dist.device, dist=dist, use_graph=True)
    else:
        from distributed_rl_amd.config import ApexConfig
        from distributed_rl_amd.engine.apex import ApexEngine
        from distributed_rl_amd.replay.device_per import DevicePER
        B, cap = 32, 4096
        cfg = ApexConfig(batch_size=B, target_update=10 ** 9)
        rep = DevicePER(cap, dist.device, batch_size=B, seed=7 + dist.rank)
        rep.fill_synthetic(cap - 64, n_action=4, seed=dist.rank)
        eng = ApexEngine(4, cfg, rep, dist.device, dist=dist, use_graph=True)
    assert dist.xgmi is not None and dist.xgmi.owns(eng.G.flat)
    p0 = eng.P.flat.clone()
    losses = []
    for _ in range(12):
        eng.step()
        losses.append(eng.stats()["loss"])
    torch.cuda.synchronize()
    mine = eng.P.flat.cpu()
    every = [None] * dist.world_size
    # both engines also agree on the batch's max IS weight: the ranks' maxima travel with the gradient all-reduce (Ape-X) /
    # in a 128-byte all-reduce under the forward pass (R2D2)
    max_w = eng.stats()["max_w"]
    td.all_gather_object(every, (float(mine.double().sum()), float(mine.abs().double().sum()), eng._graphs[0], max_w))
    ok = all(e == every[0] for e in every) and bool(torch.isfinite(mine).all()) and not torch.equal(mine, p0.cpu())
    ok = ok and max_w is not None and max_w > 0
    ok = ok and every[0][2] == "one" and all(l == l for l in losses)
    print(json.dumps({"rank": dist.rank, "ok": ok, "sums": every[0][:2], "graph": every[0][2], "loss": losses[-1],
                      "xgmi_timeout_s": dist.xgmi.timeout_s if dist.xgmi is not None else None}), flush=True)
    dist.barrier()
    dist.close()
    return 0 if ok else 1


def main() -> int:
    if "--engine" in sys.argv or "--engine-r2d2" in sys.argv:
        return engine_check("--engine-r2d2" in sys.argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--numel", type=int, default=3293184)         # Ape-X flat gradient (A = 4), 13.2 MB
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--bench", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--soak", type=int, default=0, help="N back-to-back collectives on data that changes every step, each "
                                                       "one checked (device-side checksum against the host's sum over ranks)")
    ap.add_argument("--mem", default="coarse", choices=["coarse", "fine", "uncached"])
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from distributed_rl_amd.parallel.xgmi import XgmiAllReduce
    ar = XgmiAllReduce(a.numel, dev, rank, world, n_blocks=a.blocks, mem_kind=a.mem)
    res = {"rank": rank, "world": world, "device": str(dev), "numel": a.numel, "checks": 0, "max_err": 0.0}
    ok = True
    g = torch.Generator().manual_seed(100 + rank)
    cut = (a.numel // 41) // 64 * 64                               # an engine-like (small slice | big slice) split
    cases = [(0, a.numel, 0), (cut, a.numel - cut, 0), (0, cut, 1), (64, 4, 2), (0, a.numel - 3, 3)]
    for it in range(a.iters):
        for off, n, ch in cases:
            x = torch.randn(ar.cap, generator=g)
            x[a.numel:] = 0                                        # the padding stays zero, as in the engines
            ar.flat.copy_(x)
            torch.cuda.synchronize()
            td.barrier()                                           # (the engines order this by the step itself)
            ar.all_reduce(ar.flat[off:off + n], channel=ch)
            got = ar.flat.cpu()
            want = x.clone()
            n_pad = min((n + 3) // 4 * 4, ar.cap - off)
            part = x[off:off + n_pad].clone()
            td.all_reduce(part)
            want[off:off + n_pad] = part
            err = float((got - want).abs().max())
            res["max_err"] = max(res["max_err"], err)
            res["checks"] += 1
            if not err < 1e-4 * (world ** 0.5):
                ok = False
            td.barrier()
    # back-to-back collectives on two channels inside a replayed graph (what the engines do)
    s = torch.cuda.Stream(dev)
    x = torch.full((ar.cap,), float(rank + 1))
    x[a.numel:] = 0
    stage = x.to(dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        ar.flat.copy_(stage)
        torch.cuda.synchronize()
        td.barrier()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            ar.flat.copy_(stage)
            ar.all_reduce(ar.flat[cut:a.numel], channel=0)
            ar.all_reduce(ar.flat[:cut], channel=1)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
    tot = float(sum(range(1, world + 1)))
    ok = ok and bool((ar.flat[:a.numel] == tot).all())
    res["graph_ok"] = bool((ar.flat[:a.numel] == tot).all())
    if a.soak:
        # A long run on changing data without host barriers between the collectives: step k's input is written by kernels
        # (like the backward pass) right before the collective, alternating grid sizes and channels; every rank keeps a
        # device-side record (sum, sum of squares, three probes) of each result, compared afterwards with what the host
        # computes from all ranks' inputs. A peer that read stale lines, or a flag that overtook its data, shows up here.
        base = torch.randn(a.numel, generator=g).to(dev)
        rec = torch.zeros(a.soak, 5, dtype=torch.float64, device=dev)
        probes = torch.tensor([1, a.numel // 2 + 3, a.numel - 2], device=dev)
        td.barrier()
        with torch.cuda.stream(s):
            for k in range(a.soak):
                if k % 50 == 0:
                    ar.set_blocks((16, 32, 64)[(k // 50) % 3])     # every rank switches at the same collective
                sh = (k * 9973 + 1) % a.numel
                ar.flat[:a.numel - sh].copy_(base[sh:])
                ar.flat[a.numel - sh:a.numel].copy_(base[:sh])
                ar.flat[:a.numel].mul_(1.0 + 0.0625 * ((k + rank) % 7))
                if k % 3 == 2:
                    ar.all_reduce(ar.flat[cut:a.numel], channel=0)
                    ar.all_reduce(ar.flat[:cut], channel=1)
                else:
                    ar.all_reduce(ar.flat[:a.numel], channel=k % 3)
                v = ar.flat[:a.numel].double()
                rec[k, 0], rec[k, 1] = v.sum(), (v * v).sum()
                rec[k, 2:] = v[probes]
            torch.cuda.synchronize()
        # expectation on the host: every rank's base travels once over gloo
        bases = [torch.zeros(a.numel) for _ in range(world)]
        td.all_gather(bases, base.cpu())
        worst = 0.0
        for k in range(a.soak):
            sh = (k * 9973 + 1) % a.numel
            tot = torch.zeros(a.numel, dtype=torch.float64)
            for r in range(world):
                tot += torch.roll(bases[r], -sh).double() * (1.0 + 0.0625 * ((k + r) % 7))
            want = torch.tensor([float(tot.sum()), float((tot * tot).sum()), *[float(tot[i]) for i in probes.tolist()]],
                                dtype=torch.float64)
            got = rec[k].cpu()
            err = float(((got - want).abs() / (want.abs() + 1.0)).max())
            worst = max(worst, err)
        res["soak_steps"], res["soak_worst_rel_err"] = a.soak, worst
        ok = ok and worst < 1e-4
        ar.set_blocks(a.blocks)
    if a.bench:
        td.barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            for _ in range(a.bench):
                ar.all_reduce(ar.flat[:a.numel], channel=2)
            torch.cuda.synchronize()
        res["us_per_all_reduce"] = (time.perf_counter() - t0) / a.bench * 1e6
    res["error_flag"] = ar.error()
    ok = ok and res["error_flag"] == 0
    res["ok"] = ok
    td.barrier()
    print(json.dumps(res), flush=True)
    ar.close()
    td.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
