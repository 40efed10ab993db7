"""Local launcher / supervisor (reference: run.sh:1-44).

Starts (optionally) a visdom server, the learner and N actors as child processes according to a
``config/*.conf`` file, then supervises them:

* fail-fast like ``trap 'kill $(jobs -p)'`` + ``wait -n`` (run.sh:3,44): the first child to exit ends
  the run and the rest are terminated — unless ``--restart-actors`` is given, in which case dead
  actors are respawned (actors are stateless producers);
* a heartbeat watchdog: a child whose heartbeat is older than ``--stall-seconds`` is reported and
  (with ``--kill-stalled``) terminated, which the fail-fast / restart logic then handles;
* ``learner=false`` is honoured (the reference's ``$leaner`` typo always starts one, run.sh:26).
"""
from __future__ import annotations

import argparse
import os
import signal
import subprocess
import sys
import time
import uuid
from typing import Dict, List

from ..config import parse_conf
from ..transport import shm_names

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_commands(conf, n_actors: int, server: str, run_id: str, extra_learner: List[str], extra_actor: List[str],
                   python: str = sys.executable, plan=None) -> Dict[str, List[str]]:
    """One command line per child. Learner ranks are started here, one process each (``learner``, ``learner_r1``, ...), so
    the supervisor sees every rank; devices, shards and parameter lanes come from ``plan`` (runtime/placement.py)."""
    from .placement import make_plan
    plan = plan if plan is not None else make_plan(conf, n_actors)
    redis_server = conf.redis_server or server or "localhost"       # run.sh:7-13
    visdom_server = conf.visdom_server or (server if conf.visdom else "")
    cmds: Dict[str, List[str]] = {}
    common = ["-a", conf.algorithm, "-r", redis_server, "--transport", conf.transport, "--run-id", run_id]
    if visdom_server:
        common += ["-v", visdom_server]
    if conf.visdom:
        cmds["visdom"] = [python, "-m", "visdom.server", "-logging_level", "WARNING"]       # run.sh:20-24
    n_ranks = plan.n_learners
    lane = ["--param-channel", "collective", "--param-backend", conf.param_backend]
    if conf.learner:
        for k, dev in enumerate(plan.learner_devices):
            c = [python, os.path.join(ROOT, "learner_node.py")] + common + ["--device", dev]
            if conf.actor_device:
                c += ["-d", conf.actor_device]                                               # run.sh:27-33
            if any(r[0] == k for r in plan.lanes):
                on_register = sum(1 for a in plan.actors if a.shard == k and a.channel == "register")
                c += lane + ["--register-actors", str(on_register if conf.actor else -1)]
            cmds["learner" if k == 0 else "learner_r%d" % k] = c + extra_learner
    if conf.actor:
        for a in plan.actors:                                                                # run.sh:36-42
            c = [python, os.path.join(ROOT, "actor_node.py"), "-n", str(a.no), "-t", str(n_actors)] + common + \
                ["--num-envs", str(conf.num_envs), "--learner-ranks", str(n_ranks), "--device", a.device]
            if a.channel == "collective":
                c += lane
            cmds["actor_%d" % a.no] = c + extra_actor
    return cmds


def child_envs(plan, names, port: int) -> Dict[str, Dict[str, str]]:
    """Extra environment per child: world membership (learner ranks, lane actors) and the data-parallel identity."""
    from .placement import child_env
    out: Dict[str, Dict[str, str]] = {n: {} for n in names}
    L = plan.n_learners
    for k in range(L):
        name = "learner" if k == 0 else "learner_r%d" % k
        if name in out:
            out[name].update(DRL_LEARNER_RANK=str(k), DRL_LEARNER_RANKS=str(L))
            if plan.world_size:
                out[name].update(child_env(plan, "127.0.0.1", port, k))
    for a in plan.actors:
        name = "actor_%d" % a.no
        if name in out and a.channel == "collective":
            out[name].update(child_env(plan, "127.0.0.1", port, a.world_rank))
    return out


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _free_beats(hb_table, name: str, pid: int) -> None:
    """Heartbeat rows of a reaped child (every learner rank is a child of its own)."""
    if hb_table is None:
        return
    hb_table.free_pid(pid)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="distributed_rl_amd launcher")
    ap.add_argument("--num-actors", type=int, default=4)
    ap.add_argument("--conf", type=str, default="config/all.conf")
    ap.add_argument("--server", type=str, default="")
    ap.add_argument("--run-id", type=str, default="")
    ap.add_argument("--restart-actors", action="store_true")
    ap.add_argument("--stall-seconds", type=float, default=120.0)
    ap.add_argument("--kill-stalled", action="store_true")
    ap.add_argument("--watch-period", type=float, default=5.0, help="seconds between two looks at the heartbeat table")
    ap.add_argument("--learner-args", type=str, default="", help="extra flags for learner_node.py (one string)")
    ap.add_argument("--actor-args", type=str, default="", help="extra flags for actor_node.py (one string)")
    ap.add_argument("--timeout", type=float, default=0.0, help="stop everything after this many seconds (0 = never)")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--gpus", type=int, default=-1, help="GPUs to place processes on (-1 = all visible; 0 = CPU only)")
    args = ap.parse_args(argv)

    conf = parse_conf(args.conf)
    from .placement import make_plan
    plan = make_plan(conf, args.num_actors if conf.actor else 0, args.gpus if args.gpus >= 0 else None)
    if plan.lanes and not conf.actor:
        raise SystemExit("param_channel=collective needs the lane actors in the same launcher run (one process-group world)")
    restartable = {"actor_%d" % a.no for a in plan.actors if a.channel == "register"}    # world members cannot re-join
    run_id = args.run_id or conf.run_dir or uuid.uuid4().hex[:8]
    cmds = build_commands(conf, args.num_actors, args.server, run_id, args.learner_args.split(), args.actor_args.split(),
                          plan=plan)
    print("redis server:", conf.redis_server or args.server)
    print("visdom server:", conf.visdom_server or args.server)
    print("transport:", conf.transport, "run-id:", run_id)
    for line in plan.describe():
        print("placement:", line)
    if args.dry_run:
        for k, c in cmds.items():
            print(k, " ".join(c))
        return 0

    hb_table = None
    if conf.transport == "shm":
        from .heartbeat import HeartbeatTable
        hb_table = HeartbeatTable(shm_names(run_id)[2], create=True)

    procs: Dict[str, subprocess.Popen] = {}
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONUNBUFFERED="1")

    extra_env = child_envs(plan, list(cmds), _free_port())

    def spawn(name):
        procs[name] = subprocess.Popen(cmds[name], env=dict(env, **extra_env.get(name, {})), cwd=ROOT)

    def teardown():
        for p in procs.values():
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5
        for p in procs.values():
            try:
                p.wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    stop = {"flag": False}
    restarts: Dict[str, int] = {}

    def on_signal(signum, frame):
        stop["flag"] = True

    signal.signal(signal.SIGINT, on_signal)
    signal.signal(signal.SIGTERM, on_signal)

    rc = 0
    try:
        if "visdom" in cmds:
            spawn("visdom")
            time.sleep(2)
        learners = [n for n in cmds if n == "learner" or n.startswith("learner_r")]
        for name in learners:
            spawn(name)
        if learners:
            time.sleep(1.0)       # let the learners create the shm segments actors attach to
        for name in cmds:
            if name.startswith("actor_"):
                spawn(name)
        t0 = time.time()
        last_watch = t0
        while not stop["flag"]:
            time.sleep(0.2)
            if args.timeout and time.time() - t0 > args.timeout:
                break
            for name, p in list(procs.items()):
                code = p.poll()
                if code is None:
                    continue
                _free_beats(hb_table, name, p.pid)       # a killed child could not unregister its heartbeat slot
                if name in restartable and args.restart_actors and not stop["flag"]:
                    print("[launcher] %s exited with %s, restarting" % (name, code), flush=True)
                    spawn(name)
                    restarts[name] = restarts.get(name, 0) + 1
                    continue
                print("[launcher] %s exited with %s, shutting down" % (name, code), flush=True)     # wait -n semantics
                rc = code if name.startswith("learner") or code != 0 else rc
                if name.startswith("learner") and code == 0:
                    # data-parallel learners finish together (same step count): give the other ranks a moment to return
                    # by themselves - and to write what they write at the end - before everything is torn down
                    t_grace = time.time() + 20.0
                    for other, q in procs.items():
                        if other != name and other.startswith("learner"):
                            try:
                                c2 = q.wait(max(0.1, t_grace - time.time()))
                                print("[launcher] %s exited with %s" % (other, c2), flush=True)
                                rc = rc or c2
                            except subprocess.TimeoutExpired:
                                print("[launcher] %s still running after rank exit, terminating" % other, flush=True)
                stop["flag"] = True
                break
            if hb_table is not None and time.time() - last_watch > args.watch_period:
                last_watch = time.time()
                for row in hb_table.snapshot():
                    if row["age_s"] <= args.stall_seconds or row["pid"] <= 0:
                        continue
                    # only processes THIS launcher started are ever signalled: the row is mapped back to a child of the
                    # procs table by role and ident (a stale row's pid may belong to anybody by now)
                    name = ("learner" if row["ident"] == 0 else "learner_r%d" % row["ident"]) if row["role"] == "learner" else \
                        "actor_%d" % row["ident"] if row["role"] == "actor" else ""
                    child = procs.get(name)
                    print("[launcher] %s %d (pid %d) silent for %.0f s" % (row["role"], row["ident"], row["pid"], row["age_s"]),
                          flush=True)
                    if child is None or child.poll() is not None:
                        hb_table.free_pid(row["pid"])    # nobody of ours: a leftover row, drop it
                        continue
                    if args.kill_stalled:
                        print("[launcher] terminating stalled %s (pid %d)" % (name, child.pid), flush=True)
                        try:
                            child.send_signal(signal.SIGCONT)        # a stopped process would not see the SIGTERM
                            child.terminate()
                            try:
                                child.wait(5)
                            except subprocess.TimeoutExpired:
                                child.kill()
                        except ProcessLookupError:
                            pass
    finally:
        teardown()
        if hb_table is not None:
            hb_table.close()
        if conf.transport == "shm":
            # a learner that was terminated could not unlink its segments: sweep every shard's names (best effort)
            n_ranks = max(1, int(conf.learner_gpus))
            for r in range(n_ranks):
                rid = run_id if r == 0 else "%s_r%d" % (run_id, r)
                for name in shm_names(rid):
                    try:
                        os.unlink("/dev/shm" + name)
                    except OSError:
                        pass
    return rc


if __name__ == "__main__":
    sys.exit(main())
