"""Every environment switch of docs/SWITCHES.md that no other CPU test sets is set here once, and what it promises is
checked (VERDICT r2 "Next round" #8: each name with a test that sets it). The GPU-side names are in
tests/test_switches_gpu.py; the remaining ones are set by tests/test_dist_cpu.py, test_e2e_cpu.py, test_fault_injection.py,
test_placement.py and the GPU suites (grep the name under tests/)."""
import os
import subprocess
import sys
import time

import numpy as np

from distributed_rl_amd.config import LaunchConf
from distributed_rl_amd.runtime.launcher import build_commands, child_envs
from distributed_rl_amd.runtime.placement import make_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _actor(run_id, monkeypatch, **env):
    from distributed_rl_amd import envs
    from distributed_rl_amd.config import make_config
    from distributed_rl_amd.models import DuelingDQN
    from distributed_rl_amd.runtime.actor import Actor
    from distributed_rl_amd.transport import connect
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = make_config("ape_x", actor_batch=10)
    exp, par = connect("inproc", run_id, create=True)
    venv = envs.make_vec("SyntheticCatch-v0", 2, seed=0)
    return Actor(1, venv, DuelingDQN(venv.n_action), cfg, exp, par, None, 1, seed=0), exp


def test_plane_blocks_off_sends_full_stacks(run_id, monkeypatch):
    """DRL_PLANE_BLOCKS=0: no plane log, codec kind 1 blocks with both stacks of every transition (what external producers
    and the Redis transport's consumers expect); the default sends plane pools + window positions (kind 3)."""
    actor, exp = _actor(run_id, monkeypatch, DRL_PLANE_BLOCKS="0")
    assert actor.plane_log is None
    actor.run(max_steps=40)
    blocks = exp.drain()
    assert blocks and all("state" in b and "next_state" in b and "planes" not in b for b in blocks)
    assert blocks[0]["state"].shape[1:] == (84, 84, 4) and blocks[0]["state"].dtype == np.uint8


def test_plane_blocks_default_sends_plane_pools(run_id, monkeypatch):
    actor, exp = _actor(run_id + "b", monkeypatch)
    assert actor.plane_log is not None
    actor.run(max_steps=40)
    blocks = exp.drain()
    assert blocks and all("planes" in b and "win_s" in b and "state" not in b for b in blocks)


def test_actor_report_interval(run_id, monkeypatch):
    """DRL_ACTOR_REPORT_S: seconds between two reports of an actor's rates and phase timers to the metrics sinks."""
    actor, _ = _actor(run_id, monkeypatch, DRL_ACTOR_REPORT_S="0.25")
    assert actor.report_interval == 0.25
    actor2, _ = _actor(run_id + "c", monkeypatch)
    monkeypatch.delenv("DRL_ACTOR_REPORT_S")
    actor3, _ = _actor(run_id + "d", monkeypatch)
    assert actor2.report_interval == 0.25 and actor3.report_interval == 2.0


def test_transport_run_id_connect_timeout_and_learner_ranks_from_the_environment():
    """DRL_TRANSPORT / DRL_RUN_ID are the defaults of --transport / --run-id, DRL_LEARNER_RANKS of --learner-ranks (the
    launcher sets it for its children), DRL_CONNECT_TIMEOUT bounds the wait for the learner's segments: an actor started
    for a run nobody created gives up after that long and names the segment of ITS shard (actor 2 of 2 ranks -> rank 1)."""
    env = dict(os.environ, PYTHONPATH=ROOT, DRL_TRANSPORT="shm", DRL_RUN_ID="nosuchrun_%d" % os.getpid(),
               DRL_CONNECT_TIMEOUT="0.6", DRL_LEARNER_RANKS="2")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "actor_node.py"), "-n", "2", "-t", "2", "-e", "SyntheticCatch-v0",
                        "--device", "cpu", "--quiet"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode != 0 and time.time() - t0 < 100
    assert "cannot open shm ring '/drl_nosuchrun_%d_r1_exp'" % os.getpid() in r.stderr, r.stderr[-1500:]


def test_launcher_children_get_their_rank_and_the_rendezvous():
    """DRL_LEARNER_RANK / DRL_LEARNER_RANKS / DRL_WORLD_ADDR / DRL_WORLD_PORT / DRL_WORLD_RANK / DRL_WORLD_PLAN: written by
    the launcher for its children (runtime/launcher.child_envs, runtime/placement.child_env), never by a user."""
    conf = LaunchConf(learner_gpus=2, actor_gpus="2,3", param_channel="collective", param_backend="nccl")
    plan = make_plan(conf, 4, n_gpus=4)
    cmds = build_commands(conf, 4, "", "rid", [], [], plan=plan)
    envs = child_envs(plan, list(cmds), 23456)
    assert envs["learner"]["DRL_LEARNER_RANK"] == "0" and envs["learner_r1"]["DRL_LEARNER_RANK"] == "1"
    assert {envs[n]["DRL_LEARNER_RANKS"] for n in ("learner", "learner_r1")} == {"2"}
    members = [n for n in envs if "DRL_WORLD_RANK" in envs[n]]
    assert sorted(int(envs[n]["DRL_WORLD_RANK"]) for n in members) == list(range(plan.world_size))
    assert {envs[n]["DRL_WORLD_ADDR"] for n in members} == {"127.0.0.1"}
    assert {envs[n]["DRL_WORLD_PORT"] for n in members} == {"23456"}
    assert len({envs[n]["DRL_WORLD_PLAN"] for n in members}) == 1
    a = cmds["actor_1"]
    assert a[a.index("--learner-ranks") + 1] == "2"
