#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): Ape-X learner transitions/s at batch 512 with a 2 M-transition HBM-resident
prioritized replay, on synthetic 84x84x4 uint8 frames and random-init weights; plus (in "extra") the R2D2 learner
(64 x (40+80), 100 000 sequences resident), batched GPU actor inference frames/s, the strong-scaling variant and - at
N = 1, unless ``--no-system`` - the concurrently sustained actor -> ring -> ingest -> learner rates of both algorithms.

  python bench.py --gpus N --steps K --warmup W

works as it stands for any N: with N > 1 and no rank environment the process stays GPU-free, starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`` as a
child, relays rank 0's JSON line and exits with the child's status (non-zero if any rank failed). Launched by
``torch.distributed.run`` directly (RANK / WORLD_SIZE in the environment) it is one rank.

One rank per GPU over RCCL. Weak scaling (the headline): every rank owns a replay shard of ``--replay`` transitions in
its own HBM and trains on its own 512-sample batch; gradients are summed with ONE flat all-reduce per step (direct xGMI
kernel inside the step graph when it verifies at start-up, RCCL otherwise - the line says which). A timed step is the
complete learner iteration: PER sample -> plane windows read from the store -> 3 forwards -> fused TD -> backward ->
all-reduce -> clip + RMSprop -> priority write-back (target sync included when due).

Timing: the engine is set up (replay filled, step graph captured, ``--settle`` = 1000 steps replayed so that the run is in
the state of a training job that has been going for a moment; reported as ``extra.settle_steps``), then W warm-up steps,
then exactly K steps between barrier + synchronize on both sides, max over ranks. ``extra.steady_ms_per_step`` is the same
step over a window of >= 0.5 s.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAPER_APEX_TRANSITIONS_PER_S = 9700.0     # ~19 batches/s x 512 (recalled from Horgan et al. 2018)
PAPER_R2D2_UPDATES_PER_S = 5.0
PAPER_ACTOR_FPS = 50000.0
PAPER_R2D2_ACTOR_FPS = 66000.0             # 256 actors x ~260 fps (recalled from Kapturowski et al. 2019)
_RANK_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
             "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RESTART_COUNT",
             "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE")
MIN_WINDOW_S = 0.5                         # every loop is also measured over a steady-state window at least this long


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=1000,
                    help="learner steps replayed right after the step graph has been captured, as part of setting the engine up "
                         "and BEFORE the --warmup steps: clocks, caches and the allocator reach the state of a run that has "
                         "been going for a quarter of a second (a 20-step window 5 steps after capture measured 0.206 ms per "
                         "step where every later window measured 0.199). Reported as extra.settle_steps; 0 turns it off")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--replay", type=int, default=2_000_000, help="transitions per GPU shard")
    ap.add_argument("--n-action", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the R2D2 / actor / strong-scaling side measurements")
    ap.add_argument("--r2d2-steps", type=int, default=20)
    ap.add_argument("--actor-envs", type=int, default=4096)
    ap.add_argument("--no-system", action="store_true",
                    help="skip the concurrent actors + ring + ingest + learner measurement (scripts/system_bench.py; N = 1 only)")
    ap.add_argument("--system-seconds", type=float, default=6.0)
    ap.add_argument("--check-ranks", action="store_true",
                    help="only bring up the process group (gloo without a GPU), count the ranks and print the line: "
                         "checks the self-launch path on any box")
    return ap.parse_args(argv)


# ---- parent: stays GPU-free, spawns the ranks -----------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """N > 1 without a rank environment: start the ranks as children BEFORE anything touches the GPU (no exec from a
    GPU process, no GPU context in the parent), relay rank 0's JSON line, return the launcher's exit status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:                                   # counting devices does not initialise the GPU on this image
        import torch
        n_dev = torch.cuda.device_count()
    except Exception:                      # noqa: BLE001
        n_dev = 0
    if 0 < n_dev < args.gpus:
        # rehearsal on a box with fewer GPUs than ranks: ranks share devices; RCCL refuses two ranks on one device,
        # so the process group is gloo (the JSON line says so - such a number is not a scaling measurement)
        env.setdefault("DRL_SHARE_GPU", "1")
        env.setdefault("DRL_DIST_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench: ranks exited cleanly but printed no result line\n")
        rc = 1
    return rc


# ---- one rank -------------------------------------------------------------------------------------------------------
def timed(dist, fn, steps, warmup, many=None):
    """W untimed steps, then exactly ``steps`` steps between barrier + synchronize on both sides; max over ranks.
    ``many(n)`` runs n steps in one call (ApexEngine.run_steps: several steps per graph replay, the same steps)."""
    import torch
    if many is not None:
        many(warmup)
    else:
        for _ in range(warmup):
            fn()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if many is not None:
        many(steps)
    else:
        for _ in range(steps):
            fn()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dist.max_scalar(dt)


def steady(dist, fn, chunk, min_s=MIN_WINDOW_S):
    """Seconds per call over a window of at least ``min_s`` seconds (clocks and caches settled): the same number of
    calls on every rank (the count is agreed on after each chunk), so collectives inside ``fn`` stay matched."""
    import torch
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(chunk):
            fn()
        n += chunk
        torch.cuda.synchronize()
        if dist.max_scalar(time.perf_counter() - t0) >= min_s:      # identical verdict on every rank
            break
    dist.barrier()
    torch.cuda.synchronize()
    return dist.max_scalar(time.perf_counter() - t0) / n, n


def run_system_rows(args) -> dict:
    """Whole system, concurrently, on ONE GPU: K actor processes x E envs -> shm ring -> ingest thread -> HIP learner, both
    algorithms (a child process per run: scripts/system_bench.py)."""
    out = {}
    for algo in ("ape_x", "r2d2"):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "system_bench.py"), "--algorithm", algo,
                                "--seconds", str(args.system_seconds)], capture_output=True, text=True, timeout=240, cwd=ROOT)
            last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["system_" + algo] = json.loads(last[-1]) if last else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:  # noqa: BLE001
            out["system_" + algo] = {"error": repr(e)[:200]}
    return out


def run_rank(args) -> int:
    # N = 1: the whole-system rows run FIRST, while this process has not touched the GPU yet. Run after the learner benches
    # below (as they used to be), with this process still alive, the same children measured 2 465 updates/s / 76 K env
    # steps/s where they reach ~4 000 / ~120 K on their own (profiles/system_rows_order_r3.txt): the rows are about the
    # system, not about what is left of the GPU next to a process that has just run 10 s of benchmarks on it.
    system_rows = {}
    if (not args.no_system and not args.no_extra and not args.check_ranks and args.gpus == 1
            and int(os.environ.get("WORLD_SIZE", "1")) == 1):
        system_rows = run_system_rows(args)
    import torch
    from distributed_rl_amd.config import ApexConfig, R2D2Config
    from distributed_rl_amd.models.packing import r2d2_layout
    from distributed_rl_amd.parallel.dist import Dist

    dist = Dist()
    dev = dist.device
    W = dist.world_size
    if W != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, W))
    if args.check_ranks:
        if os.environ.get("DRL_BENCH_FAIL_RANK") == str(dist.rank):      # test hook: a rank that dies must fail the run
            os._exit(3)
        ones = torch.ones(1, device=dev)
        dist.all_reduce_flat(ones)
        if dist.rank == 0:
            print(json.dumps({"metric": "rank_check", "value": int(ones.item()), "unit": "ranks", "n_gpus": W,
                              "backend": dist.backend, "device": str(dev)}), flush=True)
        dist.close()
        return 0
    dist.xgmi_reserve_numel = r2d2_layout(args.n_action).total + 64    # one communication buffer serves both engines (+ mailbox)
    from distributed_rl_amd.engine.apex import ApexEngine
    from distributed_rl_amd.ops import hip as H
    from distributed_rl_amd.replay.device_per import DevicePER

    fallbacks = []
    extra = {}
    if W > 1:       # every rank really is there: an all-reduce of ones over the process group (RCCL on a multi-GPU node)
        ones = torch.ones(1, device=dev)
        dist.all_reduce_flat(ones)
        extra["ranks_seen"] = int(ones.item())
        extra["process_group_backend"] = dist.backend
        if torch.cuda.device_count() < W:      # ranks share devices: such a number is not a scaling measurement
            extra["rehearsal"] = "%d ranks share %d GPU(s), %s process group" % (W, torch.cuda.device_count(), dist.backend)

    # size the shard to the HBM that is actually free: every frame is stored once (replay/device_per.py), ~1.3 planes of
    # 7 KB per transition -> 2 M transitions = 18 GB (as full stacks: 113 GB)
    free, total = torch.cuda.mem_get_info(dev)
    per_item = int(1.3 * 7056) + 64
    share = max(1, W // max(1, torch.cuda.device_count())) if os.environ.get("DRL_SHARE_GPU") == "1" else 1
    cap = min(args.replay, int(free * 0.8 // per_item // share))
    cfg = ApexConfig(batch_size=args.batch)
    rep = DevicePER(cap, dev, batch_size=args.batch, seed=1234 + dist.rank)
    rep.fill_synthetic(cap, n_action=args.n_action, seed=dist.rank)

    def build_apex(replay, c):
        e = ApexEngine(args.n_action, c, replay, dev, dist=dist, use_graph=not args.no_graph)
        dist.broadcast_flat(e.P.flat)                      # identical initial weights on every rank
        H.cast_bf16(e.P.flat, e.Pb.flat)
        e.sync_target()
        try:
            e.step()                                       # first step = eager warm-up + graph capture
        except Exception as ex:                            # never lose the measurement to a capture problem: eager launches
            fallbacks.append("graph->eager: %r" % (ex,))
            e.use_graph, e._graphs = False, None
            torch.cuda.synchronize()
        return e

    eng = eng_s = None
    eng = build_apex(rep, cfg)
    if args.settle > 0:                                    # same count on every rank: the collectives inside stay matched
        eng.run_steps(args.settle)
        torch.cuda.synchronize()
    extra["settle_steps"] = args.settle
    dt = timed(dist, eng.step, args.steps, args.warmup, many=eng.run_steps)
    ms = 1e3 * dt / args.steps
    updates_s = args.steps / dt
    value = updates_s * args.batch * W
    st = eng.stats()                                       # raises if a bounded inter-GPU / inter-workgroup wait timed out
    extra.update({"apex_updates_per_s": updates_s, "loss": st["loss"], "grad_norm": st["grad_norm"],
                  "replay_items_per_gpu": cap, "replay_gb_per_gpu": round(rep.bytes_per_item() * cap / 1e9, 2),
                  "graph": eng._graphs[0] if eng._graphs else "eager",
                  "steps_per_graph_replay": eng.steps_per_graph if eng._graph_k is not None else 1})
    if W > 1:
        extra["grad_all_reduce"] = "direct xGMI kernel (in graph)" if dist.xgmi is not None else "process group (%s)" % dist.backend
        if dist.all_reduce_us is not None:                 # stand-alone us per all-reduce of the gradient buffer, both paths
            extra["grad_all_reduce_us"] = dist.all_reduce_us
        extra["is_item_count_per_rank"] = eng.shard_n_is()  # N_global / world, the device scalar the IS weights read
        try:    # data-parallel sanity: every rank applied the same averaged gradients -> identical parameters everywhere
            cs = float(eng.P.flat.double().abs().sum().item())
            extra["dp_params_identical"] = bool(dist.max_scalar(cs) == -dist.max_scalar(-cs))
        except Exception as e:  # side information only
            extra["dp_params_identical"] = repr(e)[:100]

    if not args.no_extra:
        try:    # steady-state window (>= 0.5 s) of the same step, reported next to the --steps figure
            s_per, n = steady(dist, lambda: eng.run_steps(100), 1)
            s_per, n = s_per / 100, n * 100
            eng.stats()
            extra["steady_ms_per_step"] = 1e3 * s_per
            extra["steady_steps"] = n
            extra["steady_transitions_per_s"] = args.batch * W / s_per
        except Exception as e:  # noqa: BLE001
            extra["steady_error"] = repr(e)[:200]

    # ---- strong scaling: the global batch stays 512, every rank trains on 512 / W --------------------------------
    if not args.no_extra and W > 1 and args.batch % W == 0 and (args.batch // W) % 16 == 0:
        try:
            eng = None
            dist.release_grad_buffer()
            torch.cuda.empty_cache()
            bs = args.batch // W
            eng_s = build_apex(rep.with_batch(bs), ApexConfig(batch_size=bs))
            s_dt = timed(dist, eng_s.step, args.steps, args.warmup, many=eng_s.run_steps)
            eng_s.stats()
            extra["strong_scaling"] = {"global_batch": args.batch, "batch_per_gpu": bs, "ms_per_step": 1e3 * s_dt / args.steps,
                                       "transitions_per_s": args.steps / s_dt * args.batch,
                                       "graph": eng_s._graphs[0] if eng_s._graphs else "eager"}
            s_per, n = steady(dist, lambda: eng_s.run_steps(100), 1)
            extra["strong_scaling"]["steady_ms_per_step"] = 1e3 * s_per / 100
            eng = eng_s
        except Exception as e:  # noqa: BLE001
            extra["strong_scaling_error"] = repr(e)[:200]

    if not args.no_extra:
        try:    # actor inference on EVERY rank's GPU (N > 1: per-GPU minimum and the node's aggregate)
            from distributed_rl_amd.engine.actor import ActorEngine
            act = ActorEngine(args.n_action, args.actor_envs, dev, params=eng)
            a_s, n = steady(dist, act.step_host, 50)            # pinned H2D frames + inference + D2H actions
            extra["actor_env_frames_per_s"] = 4 * args.actor_envs / a_s * W      # frameskip 4; all ranks act at once
            extra["actor_env_frames_per_s_per_gpu"] = 4 * args.actor_envs / a_s  # a_s is the slowest rank's time per call
            extra["actor_inferences_per_s"] = args.actor_envs / a_s * W
            extra["actor_vs_paper"] = extra["actor_env_frames_per_s"] / PAPER_ACTOR_FPS
            g_s, n = steady(dist, act.step_synthetic, 50)       # device-resident frames (no PCIe)
            extra["actor_inferences_per_s_device_resident"] = args.actor_envs / g_s * W
            extra["actor_note"] = "inference only (H2D + forward + eps-greedy + D2H), %d GPU(s) at once; the actor LOOP " \
                                  "rate is system_*" % W
            del act
        except Exception as e:  # side metric only
            extra["actor_error"] = repr(e)[:200]
        if W > 1:
            try:    # learner -> actor GPUs: broadcast of the flat fp32 parameters, device to device over the process group
                import torch.distributed as td
                pb = {}
                if dist.backend != "nccl":
                    # rehearsal (ranks share a GPU, gloo): gloo's CUDA path stalls on back-to-back device collectives there
                    # (parallel/dist.py _device_collectives); the number would say nothing about xGMI anyway
                    raise RuntimeError("skipped: process group is %s, not RCCL" % dist.backend)
                for name, numel in (("apex_13MB", eng.P.flat.numel()), ("r2d2_32MB", r2d2_layout(args.n_action).total)):
                    buf = torch.zeros(numel, device=dev)
                    n_warm, n_it = (3, 20) if dist.backend == "nccl" else (1, 3)    # gloo (rehearsals): ~0.5 s per broadcast
                    for _ in range(n_warm):
                        td.broadcast(buf, src=0)
                    torch.cuda.synchronize()
                    dist.barrier()
                    t0 = time.perf_counter()
                    for _ in range(n_it):
                        td.broadcast(buf, src=0)
                    torch.cuda.synchronize()
                    pb[name] = round(dist.max_scalar((time.perf_counter() - t0) / n_it * 1e6), 1)
                    del buf
                pb["backend"] = dist.backend
                extra["param_broadcast_us"] = pb
            except Exception as e:  # noqa: BLE001
                extra["param_broadcast_error"] = repr(e)[:200]
        try:
            eng = rep = eng_s = None
            dist.release_grad_buffer()
            torch.cuda.empty_cache()
            from distributed_rl_amd.engine.r2d2 import R2D2Engine
            from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
            rcfg = R2D2Config()
            free, _ = torch.cuda.mem_get_info(dev)
            from distributed_rl_amd.replay.device_seq import sequences_that_fit
            # neighbouring sequences share their planes (~44 each instead of 123): 100 K paper-shape sequences are ~34 GB
            fit = sequences_that_fit(free, rcfg) // share
            scap = min(100_000, fit)
            extra["r2d2_replay_sequences_that_would_fit"] = fit
            srep = DeviceSequenceReplay(scap, dev, rcfg, seed=dist.rank)
            srep.fill_synthetic(scap, n_action=args.n_action, seed=dist.rank)
            reng = R2D2Engine(args.n_action, rcfg, srep, dev, dist=dist)
            dist.broadcast_flat(reng.P.flat)
            H.cast_bf16(reng.P.flat, reng.Pb.flat)
            reng.sync_target()
            r_settle = min(args.settle, 60)                 # (as for the headline row: capture, then a moment of running)
            for _ in range(r_settle):
                reng.step()
            extra["r2d2_settle_steps"] = r_settle
            r_dt = timed(dist, reng.step, args.r2d2_steps, 3)
            rst = reng.stats()              # the only place a timed-out wait of the persistent kernels becomes an exception
            extra["r2d2_updates_per_s"] = args.r2d2_steps / r_dt
            extra["r2d2_ms_per_step"] = 1e3 * r_dt / args.r2d2_steps
            extra["r2d2_sequence_steps_per_s"] = args.r2d2_steps / r_dt * rcfg.batch_size * rcfg.seq_len * W
            extra["r2d2_global_batch"] = rcfg.batch_size * W
            extra["r2d2_vs_paper"] = extra["r2d2_updates_per_s"] / PAPER_R2D2_UPDATES_PER_S
            extra["r2d2_replay_sequences_per_gpu"] = scap
            extra["r2d2_replay_gb_per_gpu"] = round(srep.frames.numel() / 1e9, 2)
            extra["r2d2_loss"] = rst["loss"]
            extra["r2d2_graph"] = reng._graphs[0] if reng._graphs else "eager"
            s_per, n = steady(dist, reng.step, 20)
            reng.stats()
            extra["r2d2_steady_ms_per_step"] = 1e3 * s_per
            reng = srep = None
            dist.release_grad_buffer()
            torch.cuda.empty_cache()
            from distributed_rl_amd.engine.r2d2_actor import R2D2ActorEngine
            ract = R2D2ActorEngine(args.n_action, args.actor_envs, rcfg, dev)
            ra_s, n = steady(dist, ract.step_host, 25)      # recurrent actor: frames in over PCIe, actions out
            extra["r2d2_actor_env_frames_per_s"] = 4 * args.actor_envs / ra_s * W
            extra["r2d2_actor_env_frames_per_s_per_gpu"] = 4 * args.actor_envs / ra_s
            extra["r2d2_actor_vs_paper"] = extra["r2d2_actor_env_frames_per_s"] / PAPER_R2D2_ACTOR_FPS
            del ract
        except Exception as e:
            extra["r2d2_error"] = repr(e)[:200]

    fallbacks = list(dist.fallbacks) + fallbacks
    if fallbacks:
        extra["fallbacks"] = fallbacks
    rank, world = dist.rank, W
    dist.close()

    extra.update(system_rows)                # (N = 1: measured before anything else, see the top of this function)
    if not args.no_system and not args.no_extra and world > 1:
        # whole system on the multi-GPU node, through the launcher (what ./run.sh starts): data-parallel learner ranks +
        # placed actors, (a) co-located on every GPU, (b) learner GPUs | actor GPUs with RCCL parameter broadcast lanes.
        # The other ranks have left by the time these children need the GPUs: rank 0 waits for them to exit first.
        if rank != 0:
            return 0
        torch.cuda.empty_cache()
        time.sleep(3.0)
        for mode in ("colocated", "split"):
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "system_dp_bench.py"), "--gpus", str(world),
                                    "--mode", mode, "--seconds", str(args.system_seconds)], capture_output=True, text=True,
                                   timeout=200, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in _RANK_ENV})
                last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                extra["system_dp_" + mode] = json.loads(last[-1]) if last else {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:  # noqa: BLE001
                extra["system_dp_" + mode] = {"error": repr(e)[:200]}

    if rank == 0:
        print(json.dumps({
            "metric": "apex_learner_transitions_per_sec", "value": value, "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": value / PAPER_APEX_TRANSITIONS_PER_S,
            "dtype": "bf16", "data": "synthetic",
            "config": {"model": "DuelingDQN (Nature-CNN, dueling head), Ape-X n-step double-DQN",
                       "global_batch": args.batch * world, "seq_len": 1,
                       "replay": "HBM-resident 64-ary sum-tree PER, %d transitions per GPU" % cap,
                       "parallelism": "dp%d" % world},
            "extra": extra,
        }), flush=True)
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
