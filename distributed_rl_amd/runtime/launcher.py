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
                   python: str = sys.executable) -> Dict[str, List[str]]:
    redis_server = conf.redis_server or server or "localhost"       # run.sh:7-13
    visdom_server = conf.visdom_server or (server if conf.visdom else "")
    cmds: Dict[str, List[str]] = {}
    common = ["-a", conf.algorithm, "-r", redis_server, "--transport", conf.transport, "--run-id", run_id]
    if visdom_server:
        common += ["-v", visdom_server]
    if conf.visdom:
        cmds["visdom"] = [python, "-m", "visdom.server", "-logging_level", "WARNING"]       # run.sh:20-24
    n_ranks = max(1, int(conf.learner_gpus))
    if conf.learner:
        if n_ranks > 1:      # data-parallel learners: one rank per GPU, rendezvous on loopback
            c = [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr",
                 "127.0.0.1", "--master-port", str(29400 + (abs(hash(run_id)) % 500)),
                 os.path.join(ROOT, "learner_node.py")] + common
        else:
            c = [python, os.path.join(ROOT, "learner_node.py")] + common
        if conf.actor_device:
            c += ["-d", conf.actor_device]                                                   # run.sh:27-33
        cmds["learner"] = c + extra_learner
    if conf.param_channel == "collective":
        # learner -> actors parameter sync by broadcast (RCCL between GPUs, gloo on CPU) instead of the shm register:
        # learner = rank 0, actor i = rank i of one process group (environment set per child in main())
        if n_ranks > 1:
            raise SystemExit("param_channel=collective needs learner_gpus=1 (the data-parallel learners own the default group)")
        common = common + ["--param-channel", "collective", "--param-backend", conf.param_backend]
        if "learner" in cmds:
            cmds["learner"] = cmds["learner"] + ["--param-channel", "collective", "--param-backend", conf.param_backend]
    if conf.actor:
        for i in range(1, n_actors + 1):                                                     # run.sh:36-42
            cmds["actor_%d" % i] = [python, os.path.join(ROOT, "actor_node.py"), "-n", str(i), "-t", str(n_actors)] + \
                common + ["--num-envs", str(conf.num_envs), "--learner-ranks", str(n_ranks)] + extra_actor
    return cmds


def _free_beats(hb_table, name: str, pid: int) -> None:
    """Heartbeat rows of a reaped child: its own pid, and for the learner (whose ranks may be grandchildren under
    torch.distributed.run) every row of role learner."""
    if hb_table is None:
        return
    hb_table.free_pid(pid)
    if name == "learner":
        for row in hb_table.snapshot():
            if row["role"] == "learner":
                hb_table.free_pid(row["pid"])


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
    args = ap.parse_args(argv)

    conf = parse_conf(args.conf)
    if conf.param_channel == "collective" and args.restart_actors:
        raise SystemExit("param_channel=collective and --restart-actors exclude each other (a process group cannot re-admit a rank)")
    run_id = args.run_id or conf.run_dir or uuid.uuid4().hex[:8]
    cmds = build_commands(conf, args.num_actors, args.server, run_id, args.learner_args.split(), args.actor_args.split())
    print("redis server:", conf.redis_server or args.server)
    print("visdom server:", conf.visdom_server or args.server)
    print("transport:", conf.transport, "run-id:", run_id)
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

    pc_port = str(29900 + (abs(hash(run_id)) % 500))

    def spawn(name):
        e = env
        if conf.param_channel == "collective" and name != "visdom":
            rank = 0 if name == "learner" else int(name.split("_")[1])
            e = dict(env, DRL_PC_RANK=str(rank), DRL_PC_WORLD=str(1 + args.num_actors), DRL_PC_ADDR="127.0.0.1", DRL_PC_PORT=pc_port)
        procs[name] = subprocess.Popen(cmds[name], env=e, cwd=ROOT)

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
        if "learner" in cmds:
            spawn("learner")
            time.sleep(1.0)       # let the learner create the shm segments actors attach to
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
                if name.startswith("actor_") and args.restart_actors and not stop["flag"]:
                    print("[launcher] %s exited with %s, restarting" % (name, code), flush=True)
                    spawn(name)
                    restarts[name] = restarts.get(name, 0) + 1
                    continue
                print("[launcher] %s exited with %s, shutting down" % (name, code), flush=True)     # wait -n semantics
                rc = code if name == "learner" or code != 0 else rc
                stop["flag"] = True
                break
            if hb_table is not None and time.time() - last_watch > args.watch_period:
                last_watch = time.time()
                for row in hb_table.snapshot():
                    if row["age_s"] <= args.stall_seconds or row["pid"] <= 0:
                        continue
                    # only processes THIS launcher started are ever signalled: the row is mapped back to a child of the
                    # procs table by role and ident (a stale row's pid may belong to anybody by now)
                    name = "learner" if row["role"] == "learner" else "actor_%d" % row["ident"] if row["role"] == "actor" else ""
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
