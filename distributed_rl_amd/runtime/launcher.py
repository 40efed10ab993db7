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


EXIT_GUARD = 75          # engine/learner.py: the learner's step guard fired, state saved, restart me without that path
# which switch turns off the path a fired guard names (engine/guard.py names; engine/r2d2.py, parallel/dist.py switches)
HEAL_SWITCHES = (("persistent LSTM", "DRL_PERSISTENT_LSTM"), ("persistent BPTT", "DRL_PERSISTENT_BPTT"),
                 ("xGMI", "DRL_XGMI_AR"))


def newest_checkpoint(save_dir: str) -> str:
    """``model_<t>.pth`` with the largest t in ``save_dir`` ('' if none)."""
    import glob
    import re
    best, best_t = "", -1
    for f in glob.glob(os.path.join(save_dir, "model_*.pth")):
        m = re.fullmatch(r"model_(\d+)\.pth", os.path.basename(f))
        if m and int(m.group(1)) > best_t:
            best, best_t = f, int(m.group(1))
    return best


def heal_learner(cmd: List[str], env: Dict[str, str], save_dir: str):
    """What to start in place of a learner that left with EXIT_GUARD: the same command, attached to the live segments
    (``--attach``: the actors keep running and keep their parameter versions), resumed from the checkpoint the dying learner
    wrote (or the newest older one), with every path named in its fault note switched off. Returns (cmd, env, reason)."""
    import json
    note = {}
    try:
        with open(os.path.join(save_dir, "learner_fault.json")) as fh:
            note = json.load(fh)
    except (OSError, ValueError):
        pass
    err = str(note.get("error", ""))
    off = {var: "0" for key, var in HEAL_SWITCHES if key in err}
    if not off:                                   # no note / an unknown guard: every bounded-wait path off
        off = {var: "0" for _, var in HEAL_SWITCHES}
    ckpt = note.get("checkpoint") or ""
    if not (ckpt and os.path.exists(ckpt)):
        ckpt = newest_checkpoint(save_dir)
    new = [a for a in cmd]
    if "--resume" in new:                         # replace an older --resume
        i = new.index("--resume")
        del new[i:i + 2]
    if "--attach" not in new:
        new.append("--attach")
    if ckpt:
        new += ["--resume", ckpt]
    return new, dict(env, **off), "guard fired (%s): restarting from %s with %s" % (
        err[:120] or "no fault note", ckpt or "scratch", " ".join("%s=0" % v for v in sorted(off)))


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
    ap.add_argument("--no-heal-learner", action="store_true",
                    help="a learner whose step guard fired (exit status 75) ends the run instead of being restarted from its "
                         "last checkpoint with the path that timed out switched off")
    ap.add_argument("--max-heals", type=int, default=3)
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

    def learner_save_dir() -> str:
        la = args.learner_args.split()
        d = la[la.index("--save-dir") + 1] if "--save-dir" in la and la.index("--save-dir") + 1 < len(la) else "./models"
        return d if os.path.isabs(d) else os.path.join(ROOT, d)

    heals = {"n": 0}

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
                if (code == EXIT_GUARD and name == "learner" and plan.n_learners == 1 and not plan.lanes
                        and not args.no_heal_learner and heals["n"] < args.max_heals and not stop["flag"]):
                    # self-healing: the learner discarded a step whose bounded in-kernel wait timed out, saved its state
                    # and left; a fresh process takes over the live segments, the actors never notice
                    cmds[name], new_env, why = heal_learner(cmds[name], extra_env.get(name, {}), learner_save_dir())
                    extra_env[name] = new_env
                    heals["n"] += 1
                    print("[launcher] learner %s (heal %d of %d)" % (why, heals["n"], args.max_heals), flush=True)
                    spawn(name)
                    continue
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
