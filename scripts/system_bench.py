#!/usr/bin/env python
"""Whole-system throughput on ONE GPU box: K actor processes x E envs (batched gfx950 inference, array-based n-step /
sequence assembly, initial priorities) -> shm MPSC ring -> asynchronous ingest (pinned staging + side-stream DMA) -> HBM
replay -> graph-captured learner, all running CONCURRENTLY for ``--seconds`` after a warm-up. Prints one JSON line with
the rates sustained inside that window:

  env_steps_per_s / env_frames_per_s (x4 frameskip), ingested_items_per_s, learner_updates_per_s, ring rejects, actor
  drops, parameter-version lag, per-phase wall clock of the learner loop.

The reference's equivalent is ``./run.sh N`` watched through visdom (README.md:36-49); actors, Redis and the learner's
replay thread share nothing but TCP there (ape_x/actor.py:58-102, ape_x/replay.py:24-35, ape_x/learner.py:47-49).

  python scripts/system_bench.py --algorithm ape_x --actors 3 --envs 64 --seconds 10
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _last_values(log_dir):
    """Last value of every scalar series the actors wrote (JSONL sinks)."""
    out = {}
    for p in glob.glob(os.path.join(log_dir, "actor_*.jsonl")):
        with open(p) as fh:
            for ln in fh:
                try:
                    r = json.loads(ln)
                except ValueError:
                    continue
                out[r["name"]] = r["y"]
    return out


def _score_curve(log_dir, t0, t1, buckets):
    """Mean episode score the actors reported ("Score actor_i": ape_x/actor.py:94-99 of the reference sends the same series to
    visdom) in ``buckets`` equal slices of [t0, t1]: does the running system LEARN while it is being measured?"""
    out = {}
    for p in sorted(glob.glob(os.path.join(log_dir, "actor_*.jsonl"))):
        acc = [[0.0, 0] for _ in range(buckets)]
        with open(p) as fh:
            for ln in fh:
                try:
                    r = json.loads(ln)
                except ValueError:
                    continue
                if r["name"].startswith("Score ") and t0 <= r["t"] < t1:
                    b = acc[min(buckets - 1, int((r["t"] - t0) / (t1 - t0) * buckets))]
                    b[0] += r["y"]
                    b[1] += 1
        out[os.path.basename(p)[:-6]] = [[round((i + 0.5) * (t1 - t0) / buckets, 1), round(a / n, 3) if n else None, n]
                                         for i, (a, n) in enumerate(acc)]
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--algorithm", default="ape_x", choices=["ape_x", "r2d2"])
    ap.add_argument("--actors", type=int, default=3)
    ap.add_argument("--envs", type=int, default=64, help="envs per actor process")
    ap.add_argument("--env", default="SyntheticCatch-v0")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--warmup-seconds", type=float, default=3.0)
    ap.add_argument("--replay", type=int, default=0, help="replay capacity (0: 200 000 transitions / 4 096 sequences)")
    ap.add_argument("--actor-engine", default="hip", choices=["hip", "torch"])
    ap.add_argument("--score-buckets", type=int, default=0,
                    help="also report the actors' mean episode score in this many slices of the run (is the system learning?)")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES of the LEARNER process (0: leave the default)")
    args = ap.parse_args(argv)
    if args.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)          # read when the HIP runtime starts: before torch touches the GPU

    import torch
    from distributed_rl_amd import envs
    from distributed_rl_amd.config import make_config
    from distributed_rl_amd.engine.learner import HipLearner
    from distributed_rl_amd.obs import Metrics
    from distributed_rl_amd.runtime.heartbeat import HeartbeatTable
    from distributed_rl_amd.transport import connect, shm_names

    r2 = args.algorithm == "r2d2"
    run_id = "sys" + uuid.uuid4().hex[:8]
    log_dir = os.path.join("/tmp", "drl_" + run_id)
    os.makedirs(log_dir, exist_ok=True)
    cfg = make_config(args.algorithm, start_memory_size=256 if r2 else 4096, fit_timing=1 << 30, save_timing=1 << 30)
    cap = args.replay or (4096 if r2 else 200_000)
    slot_mb = 32 if r2 else 8
    exp, par = connect("shm", run_id, create=True, n_slots=32 if r2 else 64, slot_bytes=slot_mb << 20)
    hb_table = HeartbeatTable(shm_names(run_id)[2], create=True)
    n_action = envs.n_action_of(args.env)
    learner = HipLearner(args.algorithm, n_action, cfg, exp, par, Metrics([]), cap, "cuda", publish_interval=0.05)
    learner.publish(wait=True)

    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", DRL_ACTOR_REPORT_S="1.0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if args.hw_queues:
        env.pop("GPU_MAX_HW_QUEUES", None)
    procs = []
    for i in range(1, args.actors + 1):
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "actor_node.py"), "-n", str(i), "-t", str(args.actors), "-a", args.algorithm,
             "-e", args.env, "--num-envs", str(args.envs), "--engine", args.actor_engine, "--transport", "shm", "--run-id", run_id,
             "--log-dir", log_dir, "--quiet", "--seed", str(i)], env=env, cwd=ROOT, stdout=subprocess.DEVNULL,
            stderr=open(os.path.join(log_dir, "actor_%d.err" % i), "w")))
    result = {"algorithm": args.algorithm, "actors": args.actors, "envs_per_actor": args.envs, "env": args.env,
              "actor_engine": args.actor_engine, "replay_capacity": cap}
    stop = threading.Event()
    marks = {"t0": time.time()}

    def actor_steps():
        return sum(r["progress"] for r in hb_table.snapshot() if r["role"] == "actor")

    def rss_mb():
        """Resident set of the learner (this process) and of the actor processes, MB: a soak's leak check."""
        try:
            import psutil
            return [round(psutil.Process(os.getpid()).memory_info().rss / 2 ** 20)] + \
                   [round(psutil.Process(p.pid).memory_info().rss / 2 ** 20) for p in procs if p.poll() is None]
        except Exception:  # noqa: BLE001
            return []

    def clock():
        """Marks the start and the end of the measurement window while the learner loop runs in the main thread."""
        t_begin = time.time()
        while learner.step < 5 and time.time() - t_begin < 240:      # learn-start reached + graph captured
            if any(p.poll() is not None for p in procs):             # an actor died: no point in waiting for its experience
                marks["dead"] = True
                stop.set()
                return
            time.sleep(0.05)
        time.sleep(args.warmup_seconds)
        marks["a"] = (time.time(), learner.step, learner.ingested, actor_steps(), exp.stats())
        marks["rss_a"] = rss_mb()
        halves = []
        for _ in range(2):                                   # the rate of each half of the window: does it hold?
            s0, t0_ = learner.step, time.time()
            while time.time() - t0_ < args.seconds / 2:
                time.sleep(min(30.0, max(0.0, args.seconds / 2 - (time.time() - t0_))))
                print("[system_bench] %.0f s: learner step %d, %d items ingested, replay %d, rss MB %s" % (
                    time.time() - marks["t0"], learner.step, learner.ingested, len(learner.replay), rss_mb()), file=sys.stderr, flush=True)
                if os.environ.get("SYSBENCH_DEBUG"):
                    import gc
                    import psutil
                    mi = psutil.Process(os.getpid()).memory_info()
                    print("    learner rss %d MB of which shared (shm segments, file pages) %d MB" % (mi.rss >> 20, mi.shared >> 20),
                          file=sys.stderr, flush=True)
                    print("    py objects %d, ring blocks %d, staged/retired/copying %d/%d/%d, cuda alloc %.0f MB, reserved %.0f MB" % (
                        len(gc.get_objects()), len(learner.replay._blocks), len(learner.pipe._staged), len(learner.pipe._retired),
                        len(learner.pipe._copying), torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20),
                        file=sys.stderr, flush=True)
            halves.append(round((learner.step - s0) / (time.time() - t0_)))
        marks["halves"] = halves
        marks["rss_b"] = rss_mb()
        marks["b"] = (time.time(), learner.step, learner.ingested, actor_steps(), exp.stats())
        marks["versions"] = par.version()
        stop.set()

    th = threading.Thread(target=clock, daemon=True)
    th.start()
    rc = 0
    try:
        learner.optimize_loop(save_model_dir=os.path.join(log_dir, "models"), stop_event=stop, warmup_timeout=240)
        th.join(5)
        if marks.get("dead"):
            raise RuntimeError("an actor process exited during start-up")
        learner.engine.stats()                               # raises if a bounded in-kernel wait timed out
        (ta, sa, ia, ea, ra), (tb, sb, ib, eb, rb) = marks["a"], marks["b"]
        dt = tb - ta
        last = _last_values(log_dir)
        lag = [v for k, v in last.items() if k.startswith("Param lag")]
        drops = [v for k, v in last.items() if k.startswith("Dropped blocks")]
        result.update({
            "window_s": dt,
            "learner_updates_per_s": (sb - sa) / dt,
            "learner_items_per_s": (sb - sa) * cfg.batch_size / dt,
            "ingested_items_per_s": (ib - ia) / dt,
            "env_steps_per_s": (eb - ea) / dt,
            "env_frames_per_s": 4 * (eb - ea) / dt,
            "ring_blocks_pushed": rb["pushed"] - ra["pushed"],
            "ring_rejected_pushes": rb["rejected"] - ra["rejected"],
            "ring_fill_at_end": rb["size"],
            "actor_dropped_blocks": sum(drops),
            "param_versions_published": marks["versions"],
            "param_version_lag_max": max(lag) if lag else None,
            "replay_items": len(learner.replay),
            "learner_updates_per_s_by_half": marks.get("halves"),
            "rss_mb_learner_then_actors": {"start": marks.get("rss_a"), "end": marks.get("rss_b")},
            "ingest": learner.pipe.stats() if learner.pipe is not None else "synchronous",
            "learner_phase_ms": {k: round(v["mean_ms"], 4) for k, v in learner.metrics_local.phase_report().items()},
            "actor_phase_ms": {k.split(" ", 2)[2]: round(v, 3) for k, v in last.items() if k.startswith("Phase ms actor/")
                               and k.endswith("actor_1")},
        })
        if args.score_buckets > 0:
            result["score_curve_s_mean_n"] = _score_curve(log_dir, marks["t0"], tb, args.score_buckets)
            result["learner_updates_total"] = learner.step
    except Exception as e:  # noqa: BLE001
        result["error"] = repr(e)[:300]
        rc = 1
    finally:
        stop.set()
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
        if rc:
            for i in range(1, args.actors + 1):
                try:
                    result.setdefault("actor_stderr", []).append(open(os.path.join(log_dir, "actor_%d.err" % i)).read()[-400:])
                except OSError:
                    pass
        learner.close()
        exp.close()
        par.close()
        hb_table.close()
    print(json.dumps(result), flush=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
