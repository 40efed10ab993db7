#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): Ape-X learner transitions/s at batch 512 with a 2 M-transition HBM-resident
prioritized replay, on synthetic 84x84x4 uint8 frames and random-init weights; plus (in "extra") the R2D2 learner
(64 x (40+80)) and batched GPU actor inference frames/s.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One rank per GPU over RCCL. Weak scaling: every rank owns a replay shard of ``--replay`` transitions in its own HBM
and trains on its own 512-sample batch; gradients are summed with ONE flat all-reduce per step. A timed step is
the complete learner iteration: PER sample -> frames read from the store -> 3 forwards -> fused TD ->
backward -> all-reduce -> clip + RMSprop -> priority write-back (target sync included when due).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAPER_APEX_TRANSITIONS_PER_S = 9700.0     # ~19 batches/s x 512 (recalled from Horgan et al. 2018)
PAPER_R2D2_UPDATES_PER_S = 5.0
PAPER_ACTOR_FPS = 50000.0
PAPER_R2D2_ACTOR_FPS = 66000.0             # 256 actors x ~260 fps (recalled from Kapturowski et al. 2019)


def timed(dist, fn, steps, warmup):
    import torch
    for _ in range(warmup):
        fn()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dist.max_scalar(dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--replay", type=int, default=2_000_000, help="transitions per GPU shard")
    ap.add_argument("--n-action", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the R2D2 / actor side measurements")
    ap.add_argument("--r2d2-steps", type=int, default=10)
    ap.add_argument("--actor-envs", type=int, default=4096)
    args = ap.parse_args()

    wd = float(os.environ.get("DRL_BENCH_WATCHDOG", "0"))
    if wd > 0:                               # debugging aid: dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    import torch
    from distributed_rl_amd.config import ApexConfig
    from distributed_rl_amd.parallel.dist import Dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus > 1 and world == 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (one rank per GPU)" % args.gpus)
    dist = Dist()
    dev = dist.device
    from distributed_rl_amd.engine.apex import ApexEngine
    from distributed_rl_amd.replay.device_per import DevicePER

    # size the shard to the HBM that is actually free (2 M x 56 KB = 113 GB fits an MI355X's 288 GB)
    free, total = torch.cuda.mem_get_info(dev)
    per_item = 2 * 28224 + 64
    cap = min(args.replay, int(free * 0.8 // per_item))
    cfg = ApexConfig(batch_size=args.batch)
    rep = DevicePER(cap, dev, batch_size=args.batch, seed=1234 + dist.rank)
    rep.fill_synthetic(cap, n_action=args.n_action, seed=dist.rank)
    eng = ApexEngine(args.n_action, cfg, rep, dev, dist=dist, use_graph=not args.no_graph)
    dist.broadcast_flat(eng.P.flat)
    from distributed_rl_amd.ops import hip as H
    H.cast_bf16(eng.P.flat, eng.Pb.flat)
    eng.sync_target()

    try:
        eng.step()                       # first step = eager warm-up + graph capture
    except Exception as e:               # never lose the measurement to a capture problem: fall back to eager launches
        if dist.rank == 0:
            print("graph capture failed (%r); running eagerly" % (e,), file=sys.stderr)
        eng.use_graph, eng._graphs = False, None
        torch.cuda.synchronize()
    dt = timed(dist, eng.step, args.steps, args.warmup)
    ms = 1e3 * dt / args.steps
    updates_s = args.steps / dt
    value = updates_s * args.batch * dist.world_size
    st = eng.stats()
    extra = {"apex_updates_per_s": updates_s, "loss": st["loss"], "grad_norm": st["grad_norm"],
             "replay_items_per_gpu": cap, "graph": eng._graphs[0] if eng._graphs else "eager"}
    if dist.world_size > 1:
        extra["grad_all_reduce"] = "direct xGMI kernel (in graph)" if dist.xgmi is not None else "RCCL"
        if dist.all_reduce_us is not None:                 # stand-alone us per 13 MB all-reduce measured at start-up
            extra["grad_all_reduce_us"] = dist.all_reduce_us
        try:    # data-parallel sanity: every rank applied the same averaged gradients -> identical parameters everywhere
            cs = float(eng.P.flat.double().abs().sum().item())
            extra["dp_params_identical"] = bool(dist.max_scalar(cs) == -dist.max_scalar(-cs))
        except Exception as e:  # side information only
            extra["dp_params_identical"] = repr(e)[:100]

    # side metrics only on one GPU: at N > 1 nothing after the timed region may put the headline line at risk
    if not args.no_extra and dist.world_size == 1:
        try:
            from distributed_rl_amd.engine.actor import ActorEngine
            act = ActorEngine(args.n_action, args.actor_envs, dev, params=eng)
            a_dt = timed(dist, act.step_host, 100, 10)          # pinned H2D frames + inference + D2H actions
            extra["actor_env_frames_per_s"] = 4 * args.actor_envs * 100 / a_dt * dist.world_size   # frameskip 4
            extra["actor_inferences_per_s"] = args.actor_envs * 100 / a_dt * dist.world_size
            extra["actor_vs_paper"] = extra["actor_env_frames_per_s"] / PAPER_ACTOR_FPS
            g_dt = timed(dist, act.step_synthetic, 100, 10)     # device-resident frames (no PCIe)
            extra["actor_inferences_per_s_device_resident"] = args.actor_envs * 100 / g_dt * dist.world_size
        except Exception as e:  # side metric only
            extra["actor_error"] = repr(e)[:200]
        try:
            del eng, rep
            torch.cuda.empty_cache()
            from distributed_rl_amd.config import R2D2Config
            from distributed_rl_amd.engine.r2d2 import R2D2Engine
            from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
            rcfg = R2D2Config()
            free, _ = torch.cuda.mem_get_info(dev)
            scap = min(100_000, int(free * 0.6 // (120 * 28224 + 8192)))
            srep = DeviceSequenceReplay(scap, dev, rcfg, seed=dist.rank)
            srep.fill_synthetic(scap, n_action=args.n_action, seed=dist.rank)
            reng = R2D2Engine(args.n_action, rcfg, srep, dev, dist=dist)
            r_dt = timed(dist, reng.step, args.r2d2_steps, 3)
            extra["r2d2_updates_per_s"] = args.r2d2_steps / r_dt
            extra["r2d2_sequence_steps_per_s"] = args.r2d2_steps / r_dt * rcfg.batch_size * rcfg.seq_len * dist.world_size
            extra["r2d2_vs_paper"] = extra["r2d2_updates_per_s"] / PAPER_R2D2_UPDATES_PER_S
            extra["r2d2_replay_sequences_per_gpu"] = scap
            from distributed_rl_amd.engine.r2d2_actor import R2D2ActorEngine
            del reng, srep
            torch.cuda.empty_cache()
            ract = R2D2ActorEngine(args.n_action, args.actor_envs, rcfg, dev)
            ra_dt = timed(dist, ract.step_host, 50, 5)          # recurrent actor: frames in over PCIe, actions out
            extra["r2d2_actor_env_frames_per_s"] = 4 * args.actor_envs * 50 / ra_dt * dist.world_size
            extra["r2d2_actor_vs_paper"] = extra["r2d2_actor_env_frames_per_s"] / PAPER_R2D2_ACTOR_FPS
        except Exception as e:
            extra["r2d2_error"] = repr(e)[:200]

    if dist.rank == 0:
        print(json.dumps({
            "metric": "apex_learner_transitions_per_sec", "value": value, "unit": "transitions/s",
            "n_gpus": dist.world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": value / PAPER_APEX_TRANSITIONS_PER_S,
            "dtype": "bf16", "data": "synthetic",
            "config": {"model": "DuelingDQN (Nature-CNN, dueling head), Ape-X n-step double-DQN",
                       "global_batch": args.batch * dist.world_size, "seq_len": 1,
                       "replay": "HBM-resident 64-ary sum-tree PER, %d transitions per GPU" % cap,
                       "parallelism": "dp%d" % dist.world_size},
            "extra": extra,
        }))
    dist.close()


if __name__ == "__main__":
    main()
