#!/usr/bin/env python
"""Whole-system throughput on a MULTI-GPU node, through the production path: ``runtime/launcher.py`` (what ./run.sh
starts) with a generated conf - L data-parallel HIP learner ranks (one per GPU, sharded HBM replay, gradient all-reduce
over xGMI / RCCL) and A vectorised actor processes placed by runtime/placement.py:

  --mode colocated   every GPU hosts a learner rank AND the actors that feed its shard (parameters through the shard's
                     shared-memory register; nothing but gradients crosses xGMI)
  --mode split       the first half of the GPUs learn, the second half act; parameters reach the actor GPUs by RCCL
                     broadcast from the learner rank of their shard (param_channel=collective, param_backend=nccl),
                     experience through the shards' shm rings (BASELINE.json config 3)

The processes write their JSONL metrics (obs/); after ``--seconds`` of steady state the launcher's ``--timeout`` ends the
run and this script prints ONE JSON line: learner updates/s per rank, env steps/s per actor GPU and in total, parameter
version lag, dropped blocks, actors' parameter-pull time. The reference's equivalent is ./run.sh N on a fleet watched
through visdom (README.md:36-49, aws/aws_config.yaml:1-4).

  python scripts/system_dp_bench.py --gpus 8 --mode split --seconds 10
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import tempfile
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _series(log_dir):
    """{file stem: {metric name: [values...]}} of every JSONL sink under log_dir."""
    out = {}
    for p in glob.glob(os.path.join(log_dir, "*.jsonl")):
        who = os.path.basename(p)[:-6]
        d = out.setdefault(who, {})
        with open(p) as fh:
            for ln in fh:
                try:
                    r = json.loads(ln)
                except ValueError:
                    continue
                d.setdefault(r["name"], []).append(r["y"])
    return out


def _tail_mean(v, k=4):
    v = v[-k:] if len(v) > k else v[1:] or v
    return sum(v) / max(1, len(v))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--mode", default="colocated", choices=["colocated", "split"])
    ap.add_argument("--algorithm", default="ape_x", choices=["ape_x", "r2d2"])
    ap.add_argument("--actors-per-gpu", type=int, default=2)
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--env", default="SyntheticCatch-v0")
    ap.add_argument("--seconds", type=float, default=10.0, help="steady-state window after the warm-up")
    ap.add_argument("--warmup-seconds", type=float, default=25.0, help="imports, rendezvous, replay fill, graph capture")
    ap.add_argument("--replay", type=int, default=0)
    args = ap.parse_args(argv)

    from distributed_rl_amd.config import parse_conf
    from distributed_rl_amd.runtime.placement import make_plan, visible_gpus
    r2 = args.algorithm == "r2d2"
    n_vis = visible_gpus()
    share = n_vis < args.gpus                                   # rehearsal on a smaller box: ranks share GPUs, gloo
    if args.mode == "split" and (args.gpus < 2 or share):
        print(json.dumps({"mode": args.mode, "skipped": "needs >= 2 real GPUs (visible: %d)" % n_vis}))
        return 0
    L = args.gpus if args.mode == "colocated" else args.gpus // 2
    actor_gpus = list(range(L, args.gpus)) if args.mode == "split" else []
    n_actor_gpus = len(actor_gpus) if actor_gpus else L
    n_actors = args.actors_per_gpu * n_actor_gpus
    n_actors = (n_actors + L - 1) // L * L                      # the same number of actors per shard
    tmp = tempfile.mkdtemp(prefix="drl_sysdp_")
    log_dir = os.path.join(tmp, "logs")
    os.makedirs(log_dir)
    conf_path = os.path.join(tmp, "bench.conf")
    with open(conf_path, "w") as fh:
        fh.write("actor=true\nlearner=true\nvisdom=false\nalgorithm=%s\ntransport=shm\nnum_envs=%d\nlearner_gpus=%d\n"
                 % (args.algorithm, args.envs, L))
        if actor_gpus:
            fh.write("actor_gpus=%s\nparam_channel=collective\nparam_backend=nccl\n" % ",".join(map(str, actor_gpus)))
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", DRL_ACTOR_REPORT_S="1.0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if share:
        env.update(DRL_SHARE_GPU="1", DRL_DIST_BACKEND="gloo")
    plan = make_plan(parse_conf(conf_path), n_actors, n_vis if not share else max(1, n_vis))
    cap = args.replay or (4096 if r2 else 200_000)
    start = 256 if r2 else 4096
    total_s = args.warmup_seconds + args.seconds
    run_id = "sysdp" + uuid.uuid4().hex[:6]
    cmd = [sys.executable, "-m", "distributed_rl_amd.runtime.launcher", "--num-actors", str(n_actors), "--conf", conf_path,
           "--run-id", run_id, "--timeout", str(total_s),
           "--learner-args", "--engine hip --quiet --log-dir %s -s %d --start-memory-size %d --save-timing 1000000000 "
                             "--save-dir %s -e %s --ring-slots %d --ring-slot-mb %d"
                             % (log_dir, cap, start, os.path.join(tmp, "models"), args.env, 32 if r2 else 64, 32 if r2 else 8),
           "--actor-args", "--engine hip --quiet --log-dir %s -e %s" % (log_dir, args.env)]
    res = {"mode": args.mode, "algorithm": args.algorithm, "gpus": args.gpus, "learner_ranks": L, "actors": n_actors,
           "envs_per_actor": args.envs, "actor_gpus": actor_gpus or "co-located with the learner ranks",
           "param_channel": "RCCL broadcast lanes (device to device)" if actor_gpus else "shared-memory register per shard",
           "rehearsal": "ranks share %d GPU(s), gloo" % n_vis if share else None, "replay_capacity_per_rank": cap}
    rc = 0
    try:
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=total_s + 90)
        out = p.stdout + p.stderr
        s = _series(log_dir)
        ups = {k: _tail_mean(v["Updates per second"]) for k, v in s.items() if k.startswith("learner") and "Updates per second" in v}
        ing = {k: _tail_mean(v["Ingested items per second"]) for k, v in s.items()
               if k.startswith("learner") and "Ingested items per second" in v}
        per_dev, lag, drops, pull_ms, total = {}, [], 0.0, [], 0.0
        for a in plan.actors:
            d = s.get("actor_%d" % a.no, {})
            key = "Env steps per second actor_%d" % a.no
            if key in d:
                r = _tail_mean(d[key])
                per_dev[a.device] = per_dev.get(a.device, 0.0) + r
                total += r
            lag += d.get("Param lag actor_%d" % a.no, [])[-3:]
            drops += sum(d.get("Dropped blocks actor_%d" % a.no, [])[-1:])
            pull_ms += d.get("Phase ms actor/params actor_%d" % a.no, [])[-1:]
        res.update({
            "learner_updates_per_s_per_rank": {k: round(v, 1) for k, v in sorted(ups.items())},
            "learner_updates_per_s_min": round(min(ups.values()), 1) if ups else None,
            "learner_items_per_s_total": round(sum(ups.values()) * (64 if r2 else 512), 1) if ups else None,
            "ingested_items_per_s_total": round(sum(ing.values()), 1) if ing else None,
            "env_steps_per_s_total": round(total, 1), "env_frames_per_s_total": round(4 * total, 1),
            "env_steps_per_s_per_actor_gpu": {k: round(v, 1) for k, v in sorted(per_dev.items())},
            "param_version_lag_max": max(lag) if lag else None, "actor_dropped_blocks": drops,
            "actor_param_pull_ms_mean": round(sum(pull_ms) / len(pull_ms), 3) if pull_ms else None,
            "launcher_rc": p.returncode,
        })
        if not ups or total <= 0:
            res["error"] = "no steady-state metrics (launcher output tail): " + out[-600:]
            rc = 1
    except Exception as e:  # noqa: BLE001
        res["error"] = repr(e)[:300]
        rc = 1
    print(json.dumps(res), flush=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
